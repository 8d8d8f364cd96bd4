#!/bin/bash
# round 3, GPU call 31: software-pipelined block walk (stage1 of block i+1 + its gathers in flight during stage2 of block i) at 4 and 3 waves per SIMD vs v30
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_31
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v30.so build/ab/pipe4.so build/ab/pipe3.so build/ab/v30.so build/ab/pipe4.so build/ab/pipe3.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
