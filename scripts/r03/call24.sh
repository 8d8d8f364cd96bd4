#!/bin/bash
# round 3, GPU call 24: sparse-tail share, bench.py lines (250 steps, 24-byte correspondences)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_24
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
export BTBA_BENCH_CACHE=/tmp/bench_inst.pkl
for t in 0 128 256 96 160 0 128 208; do
  BTBA_SPARSE_TAIL=$t timeout 300 python bench.py --no-cpu-baseline > "$O/bench_tail$t.json" 2> "$O/bench.err"
  python - "$O/bench_tail$t.json" $t <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print("tail", sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["frac"], round(1e3*j["roofline"]["avg_launch_ms"],2))
PY
done
for t in 0 128 256; do
  BTBA_SPARSE_TAIL=$t timeout 300 python bench.py --no-cpu-baseline --masked > "$O/bench_masked_tail$t.json" 2> "$O/bench.err"
  python - "$O/bench_masked_tail$t.json" $t <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print("masked tail", sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["frac"], round(1e3*j["roofline"]["avg_launch_ms"],2))
PY
done
