#!/bin/bash
# Round 6, GPU call 7: GPU suite on the sync-free boundary, boundary timing twice (box noise), driver-style line.
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 1800 python -m pytest tests -q -m gpu > $OUT/gputests_7.log 2>&1; tail -6 $OUT/gputests_7.log
for rep in a b; do
timeout 600 python scripts/boundary_timing.py > $OUT/boundary_timing_$rep.jsonl 2> $OUT/boundary_timing.err; python - $rep <<'PY'
import json, sys
for l in open(f"gpurun_out/r06/boundary_timing_{sys.argv[1]}.jsonl"):
    r = json.loads(l); print(r["K"], r["corr_per_pair"], r["valid_fraction"], "stateless", r["wall_ms_median"], "keyed", r["wall_ms_median_keyed"], "keyed+corr", r["wall_ms_median_keyed_frames_and_correspondences"], r["stats_ms"]["ms_cache"], r["stats_ms_keyed"]["ms_cache"], r["stats_ms_keyed"]["ms_total"])
PY
done
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style_7.json 2> $OUT/bench_driver_style_7.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_driver_style_7.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("single_instance"), {k: d["tracker_call"][k] for k in ("ms_per_call", "ms_solve", "ms_cache", "ms_upload", "ms_total")}, d["parity"]["ok"])
PY
