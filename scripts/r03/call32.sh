#!/bin/bash
# round 3, GPU call 32: timing probes of the fused sweep (wrong results by construction): 2 / 1 of the 4 tap gathers; 6 of the 21 JtJ accumulates
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_32
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v30.so build/ab/probe1.so build/ab/probe2.so build/ab/probe3.so build/ab/v30.so build/ab/probe1.so build/ab/probe2.so build/ab/probe3.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
