#!/bin/bash
# round 3, GPU call 9: pixel-loop diet (ping-pong trips, un-negated rows, no weight mask): A/B vs v21 + parity suites
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_09
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 500 python scripts/ab_libs.py build/ab/v21.so build/ab/r03g.so build/ab/v21.so build/ab/r03g.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; tail -8 $O/pytest.log
