#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r02_ransac
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_ransac.py tests/test_tracking_session.py tests/test_cpp_host.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -40 > "$OUT/pytest.log"
tail -30 "$OUT/pytest.log" | cut -c1-300
timeout 100 python tests/tools/ransac_timing.py > "$OUT/ransac_timing.jsonl" 2>&1; tail -5 "$OUT/ransac_timing.jsonl" | cut -c1-300
