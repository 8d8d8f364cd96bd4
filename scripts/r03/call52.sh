#!/bin/bash
# round 3, GPU call 52: where the dense workgroup's prologue goes (debug build with two more stamps)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_52
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
BTBA_LIB_PATH=$REPO/build/ab/protrace.so timeout 150 python scripts/prologue_trace.py > "$O/prologue_trace.json" 2> "$O/err.txt"; cat "$O/prologue_trace.json"; tail -3 "$O/err.txt"
