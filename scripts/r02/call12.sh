#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r02_boundary
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
timeout 600 python -m pytest tests/test_tracking_session.py tests/test_cpp_host.py tests/test_gpu_parity.py -m gpu -q -s -x 2>&1 | grep -v "^$" | tail -30 > "$OUT/pytest.log"
tail -12 "$OUT/pytest.log" | cut -c1-300
timeout 300 python scripts/boundary_timing.py > "$OUT/boundary_timing.jsonl" 2> "$OUT/boundary.err"; tail -3 "$OUT/boundary.err"; cut -c1-700 "$OUT/boundary_timing.jsonl"
