#!/bin/bash
# round 3, GPU call 64: k_process_depth with compile-time stencils and the interior fast path: parity tests + timing
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_64
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 300 python -m pytest tests/test_depth_processing.py -m gpu -q 2>&1 | tail -3 | tee "$O/pytest.txt"
timeout 200 python tests/tools/image_timing.py > "$O/image_timing.json" 2> "$O/err.txt"; cat "$O/image_timing.json" | cut -c1-700; tail -2 "$O/err.txt"
