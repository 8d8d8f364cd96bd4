#!/bin/bash
# round 3, GPU call 39: k_system_solve: sparse and dense partial sums reduced at the same time by disjoint wave groups
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_39
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v31c.so build/ab/v31d.so build/ab/v31c.so build/ab/v31d.so build/ab/v31c.so build/ab/v31d.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
