#!/bin/bash
# Round 6, GPU call 11: the solve's update phase with 1-ulp division / square root (BTBA_SOLVE_FAST_SE3) against IEEE: timing, pose distance, the GPU suite.
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python scripts/ab_libs.py build/ab/r06_ieee.so build/ab/r06_fast.so build/ab/r06_ieee.so build/ab/r06_fast.so > $OUT/ab_fastse3.jsonl 2> $OUT/ab_fastse3.err
AB_B=1 timeout 600 python scripts/ab_libs.py build/ab/r06_ieee.so build/ab/r06_fast.so build/ab/r06_ieee.so build/ab/r06_fast.so > $OUT/ab_fastse3_b1.jsonl 2>> $OUT/ab_fastse3.err
python - <<'PY'
import json
for f in ("ab_fastse3", "ab_fastse3_b1"):
    for l in open(f"gpurun_out/r06/{f}.jsonl"):
        if l.startswith("{"):
            r = json.loads(l); print(f, r["lib"], r["full"]["ms_per_step"], r["full"]["sweep_us"], r["full"]["solve_us"], r["full"]["checksum"], "| masked", r["masked"]["ms_per_step"], r["masked"]["sweep_us"], r["masked"]["solve_us"], r["masked"]["checksum"])
        else: print(l[:300])
PY
timeout 300 python scripts/r06/determinism.py build/ab/r06_fast.so > $OUT/determinism4.jsonl 2>&1; python -c "
import json
for l in open('gpurun_out/r06/determinism4.jsonl'):
    if l.startswith('{'):
        r=json.loads(l); print(r['lib'], r['full']['deterministic'], r['masked']['deterministic'])"
timeout 1800 python -m pytest tests -q -m gpu > $OUT/gputests_11.log 2>&1; tail -8 $OUT/gputests_11.log
