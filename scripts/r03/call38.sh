#!/bin/bash
# round 3, GPU call 38: the fused sweep compiled for 5 waves per SIMD (96 VGPRs)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_38
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v31c.so build/ab/v31c_w5.so build/ab/v31c.so build/ab/v31c_w5.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
