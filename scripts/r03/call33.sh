#!/bin/bash
# round 3, GPU call 33: unconditional (clamped) stream prefetch in the block walk: at the top of the trip / behind the tap gathers
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_33
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v30.so build/ab/pf_top.so build/ab/pf_mid.so build/ab/v30.so build/ab/pf_top.so build/ab/pf_mid.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
