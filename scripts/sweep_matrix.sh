#!/bin/bash
# A/B matrix of fused-sweep schedule switches at the bench workloads (GPU box): one bench.py line per (config, environment), instances generated once per config.
#   scripts/sweep_matrix.sh <out.jsonl> <config> "<ENV1=..> <ENV2=..>" "<...>" ...      (each quoted argument = the environment of one run; "" = defaults)
OUT=$1; CFG=$2; shift 2
export BTBA_BENCH_CACHE=/tmp/sweep_matrix_$CFG.npz
for envs in "$@"; do
  line=$(env $envs python bench.py --config $CFG --no-cpu-baseline --no-tracker-call --steps 30 --warmup 5 2>/dev/null | tail -1)
  python - "$envs" "$line" >> $OUT <<'PY'
import json, sys
d = json.loads(sys.argv[2])
k = d.get("kernels_ms_per_step", {})
print(json.dumps({"config": d["config"]["workload"].split(":")[0], "env": sys.argv[1], "value": d["value"], "ms_per_step": d["ms_per_step"], "sweep_us": round(1e3 * d["roofline"]["avg_launch_ms"], 2) if "roofline" in d else None,
                  "solve_ms_per_step": k.get("system_solve"), "tiles": d["config"]["dense_tiles"], "executed": (d.get("roofline", {}).get("executed") or {}).get("frac")}))
PY
done
tail -n $# $OUT
