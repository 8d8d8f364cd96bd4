#!/bin/bash
# round 3, GPU call 36: per-trip phase stamps of the block walk (wave 0 of every dense workgroup)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_36
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
BTBA_LIB_PATH=$REPO/build/ab/triptrace.so timeout 400 python scripts/trip_trace.py > "$O/trip_trace.json" 2> "$O/err.txt"; cat "$O/trip_trace.json"; tail -5 "$O/err.txt"
