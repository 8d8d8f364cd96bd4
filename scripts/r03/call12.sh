#!/bin/bash
# round 3, GPU call 12: 24-byte correspondences -- new test, session tests (keyed pool), parity suite
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_12
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_tracking_session.py tests/test_cpp_host.py tests/test_cpp_bundler.py -m gpu -q -x 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; tail -25 $O/pytest.log
