#!/bin/bash
# round 3, GPU call 65: the profile set of record for build v33 (= v32 + the faster depth pre-processing kernel; solver kernels unchanged) + the GPU test suite
# (default incl. CPU baseline, masked, c2, c4, latency), boundary / large-window / RANSAC timings
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_65
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
bash scripts/profile_bench.sh r03_c3 > "$O/prof_c3.log" 2>&1
bash scripts/profile_bench.sh r03_masked --masked > "$O/prof_masked.log" 2>&1
bash scripts/profile_bench.sh r03_c2 --config c2 > "$O/prof_c2.log" 2>&1
bash scripts/profile_bench.sh r03_c4 --config c4 --steps 40 > "$O/prof_c4.log" 2>&1
tail -2 "$O"/prof_*.log
timeout 400 python bench.py --latency > "$O/bench_default.json" 2> "$O/bench_default.err"
for v in "masked:--masked" "masked_corr24:--masked --corr24" "c2:--config c2" "c4:--config c4 --steps 40" "entryj:--entryj"; do
  name=${v%%:*}; args=${v#*:}
  timeout 400 python bench.py --no-cpu-baseline $args > "$O/bench_$name.json" 2> "$O/bench_$name.err"
done
for f in "$O"/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j.get("roofline",{})
    print(sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], r.get("bound"), r.get("frac"), r.get("avg_launch_ms"), "traffic", r.get("traffic"), j.get("single_instance"), (j.get("cpu_baseline") or {}).get("value"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
timeout 300 python scripts/boundary_timing.py > "$O/boundary_timing_v33.jsonl" 2> "$O/boundary.err"; tail -5 "$O/boundary_timing_v33.jsonl" | cut -c1-400
timeout 300 python scripts/large_window_timing.py > "$O/large_window_timing_v33.jsonl" 2> "$O/large.err"; cat "$O/large_window_timing_v33.jsonl" | cut -c1-300
timeout 200 python tests/tools/ransac_timing.py > "$O/ransac_timing_v33.jsonl" 2> "$O/ransac.err"; tail -3 "$O/ransac_timing_v33.jsonl" | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee "$O/pytest_gpu.txt"
timeout 200 python tests/tools/image_timing.py > "$O/image_timing_v33.json" 2>> "$O/err_image.txt"; cat "$O/image_timing_v33.json" | cut -c1-600
