#!/bin/bash
# round 3, GPU call 26: masked line again (interleave period fix) + full GPU suite on the new defaults
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_26
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
for v in "masked:--masked" "default:"; do
  name=${v%%:*}; args=${v#*:}
  timeout 400 python bench.py --no-cpu-baseline $args > "$O/bench_$name.json" 2> "$O/bench.err"
  python - "$O/bench_$name.json" $name <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["bound"], j["roofline"]["frac"], round(1e3*j["roofline"]["avg_launch_ms"],2))
PY
done
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; tail -4 $O/pytest.log
