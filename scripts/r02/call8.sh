#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r02_full2
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
timeout 300 python bench.py --latency > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 600 "$OUT/bench.err"; cut -c1-1500 "$OUT/bench.json"
timeout 120 python bench.py --masked --no-cpu-baseline > "$OUT/bench_masked.json" 2>> "$OUT/bench.err"
timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -70 > "$OUT/pytest.log"
tail -40 "$OUT/pytest.log"
