#!/bin/bash
# round 3, GPU call 21: full GPU suite (tightened tests, self-spawning bench, atomic mode, 24-byte correspondences)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_21
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; grep -n "passed\|failed\|FAILED\|BA call\|window .* (poses far off)\|atomic vs" $O/pytest.log | tail -25
cp gpurun_out/session_parity_per_box.jsonl "$O/" 2>/dev/null
