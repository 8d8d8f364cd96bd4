#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r02_tests3
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > "$OUT/pytest_full.log"
grep -n "passed\|failed\|FAILED\|Error\|first differing\|windows take\|random windows\|BA call\|HIP vs the ref\|c4:\|TARGET\|EXPLICIT" "$OUT/pytest_full.log" | cut -c1-400 | tail -50
