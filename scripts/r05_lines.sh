#!/bin/bash
# Lines of record of round 5 (GPU box): the GPU test suite, the bench lines, the boundary timing.  Everything under gpurun_out/r05_final/.
OUT=gpurun_out/r05_final; mkdir -p $OUT/bench_lines
timeout 900 python -m pytest tests -q -m gpu > $OUT/bench_lines/gputests.log 2>&1; tail -3 $OUT/bench_lines/gputests.log
timeout 300 python bench.py > $OUT/bench_lines/bench_default.json 2> $OUT/bench_lines/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_lines/bench_driver_style.json 2> $OUT/bench_lines/bench_driver_style.err
timeout 300 python bench.py --masked --no-cpu-baseline --no-tracker-call > $OUT/bench_lines/bench_masked.json 2>/dev/null
timeout 300 python bench.py --config c2 --no-cpu-baseline --no-tracker-call > $OUT/bench_lines/bench_c2.json 2>/dev/null
timeout 400 python bench.py --config c4 --no-cpu-baseline --no-tracker-call > $OUT/bench_lines/bench_c4.json 2>/dev/null
timeout 300 python bench.py --latency --no-cpu-baseline --no-tracker-call > $OUT/bench_lines/bench_latency.json 2>/dev/null
timeout 300 python bench.py --masked --latency --no-cpu-baseline --no-tracker-call --instances 1 --distinct 1 > $OUT/bench_lines/bench_masked_single.json 2>/dev/null
timeout 300 python scripts/boundary_timing.py > $OUT/boundary_timing.jsonl 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_final/bench_lines/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("value_incl_pack"), (d.get("roofline") or {}).get("frac"), ((d.get("roofline") or {}).get("executed") or {}).get("frac"), (d.get("kernels_ms_per_step") or {}).get("system_solve"), d.get("single_instance"), (d.get("parity") or {}).get("worst_rot"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -4 $OUT/boundary_timing.jsonl | cut -c1-400
