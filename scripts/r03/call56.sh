#!/bin/bash
# round 3, GPU call 56: the default bench line once more (box-to-box spread of the headline)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_56
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
n=$(ls "$O" | wc -l)
timeout 400 python bench.py --no-cpu-baseline > "$O/bench_$n.json" 2> "$O/bench_$n.err"
python - "$O/bench_$n.json" <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]
print(j["value"], j["ms_per_step"], r["frac"], r["avg_launch_ms"], r.get("hbm_algorithmic",{}).get("peak_measured_copy_GBps"))
PY
