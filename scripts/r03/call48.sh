#!/bin/bash
# round 3, GPU call 48: k_system_solve phase B1 from per-element records (host table, loaded at kernel start)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_48
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v31.so build/ab/solve_t1.so build/ab/v31.so build/ab/solve_t1.so build/ab/v31.so build/ab/solve_t1.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
