#!/bin/bash
# round 3, GPU call 17: XCD-aware instance remap of k_system_solve: A/B
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_17
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 500 python scripts/ab_libs.py build/ab/r03n_base.so build/ab/r03m.so build/ab/r03n_base.so build/ab/r03m.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
