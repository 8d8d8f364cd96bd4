#!/bin/bash
# round 3, GPU call 49: workgroup timeline of the v31 fused sweep (scripts/wg_trace.py)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_49
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
BTBA_LIB_PATH=$REPO/build/ab/trace.so timeout 400 python scripts/wg_trace.py > "$O/wg_trace.json" 2> "$O/err.txt"; cat "$O/wg_trace.json"; tail -3 "$O/err.txt"
