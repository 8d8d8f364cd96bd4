#!/bin/bash
# Round 6, GPU call 4: the fixed diet variants (determinism, timing), the GPU suite on the product build, the 120-window fuzz against the reference and its self-spread.
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python scripts/r06/determinism.py build/ab/r06_base.so build/ab/r06_safe.so build/ab/r06_pose_fixed.so build/ab/r06_all_fixed.so build/ab/r06_vgpr5_fixed.so > $OUT/determinism2.jsonl 2> $OUT/determinism2.err; python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r06/determinism2.jsonl") if l.startswith("{")]
base = rows[0]
for r in rows:
    print(r["lib"], {t: (r[t]["deterministic"], r[t]["per_instance"] == base[t]["per_instance"]) for t in ("full", "masked")})
PY
timeout 900 python scripts/ab_libs.py build/ab/r06_base.so build/ab/r06_safe.so build/ab/r06_lutonly.so build/ab/r06_pose_fixed.so build/ab/r06_all_fixed.so build/ab/r06_vgpr5_fixed.so build/ab/r06_base5.so build/ab/r06_base.so build/ab/r06_safe.so > $OUT/ab_diet3.jsonl 2> $OUT/ab_diet3.err; python - <<'PY'
import json
for l in open("gpurun_out/r06/ab_diet3.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r["lib"], r["full"]["ms_per_step"], r["full"]["sweep_us"], r["full"]["checksum"], "| masked", r["masked"]["ms_per_step"], r["masked"]["sweep_us"], r["masked"]["checksum"])
    else: print(l[:300])
PY
timeout 1800 python -m pytest tests -q -m gpu > $OUT/gputests_4.log 2>&1; tail -15 $OUT/gputests_4.log
timeout 1500 python tests/tools/fuzz_parity.py 120 > $OUT/fuzz_parity_120.jsonl 2> $OUT/fuzz.err; tail -1 $OUT/fuzz_parity_120.jsonl
