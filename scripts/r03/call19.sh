#!/bin/bash
# round 3, GPU call 19: k_system_solve: two-row PCG instantiation (o), + sincos (p): A/B (checksums = bit-identity) and phase stamps
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_19
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 500 python scripts/ab_libs.py build/ab/r03n_base.so build/ab/r03o.so build/ab/r03p.so build/ab/r03n_base.so build/ab/r03o.so build/ab/r03p.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
timeout 300 python scripts/sys_clocks.py > "$O/sys_clocks.txt" 2> "$O/sys_clocks.err"; cat "$O/sys_clocks.txt"
