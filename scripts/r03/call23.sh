#!/bin/bash
# round 3, GPU call 23: share of the sparse items held back to the end of the fused launch
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_23
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 900 python scripts/ab_libs.py build/ab/r03r.so build/ab/r03r.so:BTBA_SPARSE_TAIL=64 build/ab/r03r.so:BTBA_SPARSE_TAIL=128 build/ab/r03r.so:BTBA_SPARSE_TAIL=192 build/ab/r03r.so:BTBA_SPARSE_TAIL=256 build/ab/r03r.so build/ab/r03r.so:BTBA_SPARSE_TAIL=128 > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
