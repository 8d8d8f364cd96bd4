#!/bin/bash
# round 3, GPU call 8: workgroup timeline of the fused sweep with the one-round prologue; rest of the GPU suite
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_08
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
BTBA_LIB_PATH=build/ab/trace.so timeout 300 python scripts/wg_trace.py > "$O/trace.json" 2> "$O/trace.err"; cat "$O/trace.json"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; tail -8 $O/pytest.log
