#!/bin/bash
# Round 6, GPU call 5: the sync-free keyed frame cache + fused frame-cache launch + upload on its own stream: GPU suite, boundary timing, a driver-style line.
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 1800 python -m pytest tests -q -m gpu > $OUT/gputests_5.log 2>&1; tail -12 $OUT/gputests_5.log
timeout 600 python scripts/boundary_timing.py > $OUT/boundary_timing.jsonl 2> $OUT/boundary_timing.err; python - <<'PY'
import json
for l in open("gpurun_out/r06/boundary_timing.jsonl"):
    r = json.loads(l); print(r["K"], r["corr_per_pair"], r["valid_fraction"], "stateless", r["wall_ms_median"], "keyed", r["wall_ms_median_keyed"], "keyed+corr", r["wall_ms_median_keyed_frames_and_correspondences"], r["stats_ms"], r["stats_ms_keyed"])
PY
tail -3 $OUT/boundary_timing.err
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style_5.json 2> $OUT/bench_driver_style_5.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_driver_style_5.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("single_instance"), d.get("tracker_call"), d.get("parity"))
PY
