#!/bin/bash
# round 3, GPU call 66: the bench lines of record (they now find their workload's counter record in profiles/sweep_counters.json)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_66
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 400 python bench.py --latency > "$O/bench_default.json" 2> "$O/bench_default.err"
for v in "masked:--masked" "c2:--config c2" "c4:--config c4 --steps 40"; do
  name=${v%%:*}; args=${v#*:}
  timeout 400 python bench.py --no-cpu-baseline $args > "$O/bench_$name.json" 2> "$O/bench_$name.err"
done
for f in "$O"/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j.get("roofline",{})
    print(sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], r.get("bound"), r.get("frac"), r.get("avg_launch_ms"), "traffic", r.get("traffic"), (r.get("valu_issue") or {}).get("busy_frac"), j.get("single_instance"), (j.get("cpu_baseline") or {}).get("value"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
