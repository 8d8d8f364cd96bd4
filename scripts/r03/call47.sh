#!/bin/bash
# round 3, GPU call 47: k_system_solve assembly with LDS reads batched in rounds of four (off-diagonal blocks, diagonal blocks, right-hand side)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_47
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v31.so build/ab/solve_b.so build/ab/v31.so build/ab/solve_b.so build/ab/v31.so build/ab/solve_b.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
