#!/bin/bash
# round 3, GPU call 5: persistent fused sweep with 32 cursors per XCD + stealing: A/B and item timeline
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_05
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 500 python scripts/ab_libs.py build/ab/v21.so build/ab/r03d.so:BTBA_NO_PERSISTENT=1 build/ab/r03d.so build/ab/r03d.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
BTBA_LIB_PATH=build/ab/trace.so timeout 300 python scripts/wg_trace.py > "$O/trace_persist.json" 2> "$O/trace_persist.err"; cat "$O/trace_persist.json"
