#!/bin/bash
# Round 6, GPU call 6: the fused frame-cache kernel with deeper load batches: boundary tests + timing.
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -k "cache or boundary or session or tracker or abi or entry or depth or golden or cpp" > $OUT/gputests_6.log 2>&1; tail -5 $OUT/gputests_6.log
timeout 600 python scripts/boundary_timing.py > $OUT/boundary_timing.jsonl 2> $OUT/boundary_timing.err; python - <<'PY'
import json
for l in open("gpurun_out/r06/boundary_timing.jsonl"):
    r = json.loads(l); print(r["K"], r["corr_per_pair"], r["valid_fraction"], "stateless", r["wall_ms_median"], "keyed", r["wall_ms_median_keyed"], "keyed+corr", r["wall_ms_median_keyed_frames_and_correspondences"], r["stats_ms"]["ms_cache"], r["stats_ms_keyed"]["ms_cache"], r["stats_ms_keyed"]["ms_total"])
PY
