#!/bin/bash
# round 3, GPU call 43: timing probe (wrong results): taps out of a per-wave 16 x 16 texel LDS window refilled by four LDS-direct loads per lane
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_43
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v31.so build/ab/probe_lds.so build/ab/v31.so build/ab/probe_lds.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
