#!/bin/bash
# Round 6, final check on the committed tree: smoke, the GPU suite, the driver's bench command.
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
timeout 1800 python -m pytest tests -q -m gpu > $OUT/gputests_final.log 2>&1; tail -4 $OUT/gputests_final.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_final.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], "roofline", r["frac"], r["executed"]["frac"], "traffic", r["traffic"], r["valu_issue"]["busy_frac"] if r["valu_issue"] else None, "single", d["single_instance"]["ms_per_solve"], "tracker", d["tracker_call"]["ms_per_call"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "parity", d["parity"]["ok"], d["parity"]["worst_rot"])
PY
