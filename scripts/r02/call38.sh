#!/bin/bash
# round 2, GPU call 38: the two test files that changed after call 37 (HIP default RANSAC call vs the reference's whole ransacMultiPairGPU; C++ Bundler session)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r02_38
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 300 python -m pytest tests/test_gpu_ransac.py tests/test_cpp_bundler.py tests/test_tracking_session.py -m gpu -q -s 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; tail -6 $O/pytest.log
