#!/bin/bash
# round 3, GPU call 44: BTBA_FLAG_OVERLAP (instance groups on two streams) re-measured on build v31, full and masked frames
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_44
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v31.so:AB_NO_TIMING=1 build/ab/v31.so:AB_NO_TIMING=1:AB_FLAGS=32 build/ab/v31.so:AB_NO_TIMING=1:AB_FLAGS=32:BTBA_GROUPS=4 build/ab/v31.so:AB_NO_TIMING=1:AB_FLAGS=32:BTBA_GROUP_PRIO=e build/ab/v31.so:AB_NO_TIMING=1 build/ab/v31.so:AB_NO_TIMING=1:AB_FLAGS=32 > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
