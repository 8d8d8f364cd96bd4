#!/bin/bash
# round 3, GPU call 54: the randomised-window test + entry-order test
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_54
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_entry.py -m gpu -q -x -k "randomised or build_then_smoke" 2>&1 | tail -8 | tee "$O/pytest.txt"
timeout 250 python tests/tools/fuzz_parity.py 40 > "$O/fuzz.jsonl" 2> "$O/err.txt"; tail -1 "$O/fuzz.jsonl"
