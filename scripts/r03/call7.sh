#!/bin/bash
# round 3, GPU call 7: dense workgroups that compute their item and read the poses with scalar loads (no k_pair_setup, no work-table round trip): A/B vs v21, full GPU suite
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_07
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 500 python scripts/ab_libs.py build/ab/v21.so build/ab/r03f.so build/ab/v21.so build/ab/r03f.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; tail -8 $O/pytest.log
