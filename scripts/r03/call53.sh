#!/bin/bash
# round 3, GPU call 53: randomised parity sweep of the boundary against the oracle (tests/tools/fuzz_parity.py)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_53
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 280 python tests/tools/fuzz_parity.py 40 > "$O/fuzz.jsonl" 2> "$O/err.txt"; grep -v "\"rot\": [0-9.]*e-0[5-9]" "$O/fuzz.jsonl" | cut -c1-700; tail -5 "$O/err.txt"
