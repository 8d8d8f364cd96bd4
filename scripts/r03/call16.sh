#!/bin/bash
# round 3, GPU call 16: three 24-byte entries in flight per lane: bench lines
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_16
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
for v in "default:" "entryj:--entryj" "masked:--masked" "masked_entryj:--masked --entryj" "c2:--config c2" "c2_entryj:--config c2 --entryj"; do
  name=${v%%:*}; args=${v#*:}
  timeout 400 python bench.py --no-cpu-baseline $args > "$O/bench_$name.json" 2> "$O/bench_$name.err"
  python - "$O/bench_$name.json" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1]))
    k=j.get("kernels_ms_per_step",{})
    print(sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], "roofline", j.get("roofline",{}).get("bound"), j.get("roofline",{}).get("frac"), j.get("roofline",{}).get("avg_launch_ms"), "solve", k.get("system_solve"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
