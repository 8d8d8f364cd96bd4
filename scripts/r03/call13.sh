#!/bin/bash
# round 3, GPU call 13: bench lines with the 24-byte resident correspondences: default, masked, c2, c4, and EntryJ for comparison
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_13
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
for v in "default:--latency" "entryj:--entryj" "masked:--masked" "masked_entryj:--masked --entryj" "c2:--config c2" "c2_entryj:--config c2 --entryj" "c4:--config c4 --steps 40"; do
  name=${v%%:*}; args=${v#*:}
  timeout 400 python bench.py --no-cpu-baseline $args > "$O/bench_$name.json" 2> "$O/bench_$name.err"
  python - "$O/bench_$name.json" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1]))
    k=j.get("kernels_ms_per_step",{})
    print(sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], "roofline", j.get("roofline",{}).get("bound"), j.get("roofline",{}).get("frac"), j.get("roofline",{}).get("avg_launch_ms"), "solve", k.get("system_solve"), j.get("single_instance"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
BTBA_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --instances 4 --distinct 2 --no-cpu-baseline > "$O/bench_gpus2.json" 2> "$O/bench_gpus2.err"; tail -c 600 "$O/bench_gpus2.json"; tail -2 "$O/bench_gpus2.err"
timeout 60 python bench.py --gpus 2 --steps 3 --warmup 1 --instances 4 --distinct 2 --no-cpu-baseline > "$O/bench_gpus2_refused.json" 2> "$O/bench_gpus2_refused.err"; echo "rc=$?"; tail -1 "$O/bench_gpus2_refused.err"
