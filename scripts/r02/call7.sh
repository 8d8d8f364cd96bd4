#!/bin/bash
# round-2 GPU call: the whole gpu test suite + the default bench line + the profile passes of the current kernels
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r02_full1
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -60 > "$OUT/pytest.log"
tail -30 "$OUT/pytest.log"
timeout 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 2500 "$OUT/bench.json"
timeout 900 bash scripts/profile_bench.sh r02a > "$OUT/profile.log" 2>&1
