#!/bin/bash
# round 3, GPU call 3: persistent fused sweep with the cursor read one item ahead
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_03
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 500 python scripts/ab_libs.py build/ab/v21.so build/ab/r03c.so:BTBA_NO_PERSISTENT=1 build/ab/r03c.so build/ab/r03c.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
