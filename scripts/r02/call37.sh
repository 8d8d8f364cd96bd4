#!/bin/bash
# round 2, GPU call 37: the full GPU suite (call 36 stopped at its first test)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r02_37
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 600 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > "$O/pytest_full.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest_full.log; tail -4 $O/pytest_full.log
