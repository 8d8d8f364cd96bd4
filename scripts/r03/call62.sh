#!/bin/bash
# round 3, GPU call 62: fuzz case 118 (two frames, no valid correspondence, fully valid frames) against the fp64 restatement
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_62
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 500 python tests/tools/fuzz_parity.py 120 118 > "$O/case118.jsonl" 2> "$O/err.txt"; cat "$O/case118.jsonl" | cut -c1-6000; tail -3 "$O/err.txt"
