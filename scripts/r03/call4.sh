#!/bin/bash
# round 3, GPU call 4: workgroup / item timelines of the fused sweep, one workgroup per item vs persistent
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_04
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
BTBA_LIB_PATH=build/ab/trace.so BTBA_NO_PERSISTENT=1 timeout 300 python scripts/wg_trace.py > "$O/trace_items.json" 2> "$O/trace_items.err"; cat "$O/trace_items.json"
BTBA_LIB_PATH=build/ab/trace.so timeout 300 python scripts/wg_trace.py > "$O/trace_persist.json" 2> "$O/trace_persist.err"; cat "$O/trace_persist.json"; tail -3 "$O/trace_persist.err"
