#!/bin/bash
# round 3, GPU call 22: the tightened dead-block test
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_22
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; grep -n "passed\|failed\|FAILED\|dead-block\|^E " $O/pytest.log | tail -15
