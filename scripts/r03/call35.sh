#!/bin/bash
# round 3, GPU call 37: the fused sweep compiled for 7 (72 VGPRs, 11 spilled outside the loops) and 8 (64 VGPRs, 42 spilled) waves per SIMD
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_37
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v31c.so build/ab/v31c_w7.so build/ab/v31c_w8.so build/ab/v31c.so build/ab/v31c_w7.so build/ab/v31c_w8.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
