#!/bin/bash
# round 2, GPU call 42: the golden-fixture test of test_gpu_parity.py after the XORWOW table moved out of its glob (the other 74 GPU tests passed in call 41b on this build)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r02_42
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; tail -3 $O/pytest.log
