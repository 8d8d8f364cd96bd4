#!/bin/bash
# round 3, GPU call 18: phase stamps of k_system_solve (trace mode)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_18
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 300 python scripts/sys_clocks.py > "$O/sys_clocks.txt" 2> "$O/sys_clocks.err"; cat "$O/sys_clocks.txt"; tail -3 "$O/sys_clocks.err"
