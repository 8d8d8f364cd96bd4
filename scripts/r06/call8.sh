#!/bin/bash
# Round 6, GPU call 8: the pair records written by the solve (verdict item 4b): on / off A/B in one library (env BTBA_NO_PAIR_RECORDS), determinism, workgroup timeline, tests.
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python scripts/r06/determinism.py build/ab/r06_rec.so build/ab/r06_rec.so:BTBA_NO_PAIR_RECORDS=1 build/ab/r06_safe.so > $OUT/determinism3.jsonl 2> $OUT/determinism3.err; python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r06/determinism3.jsonl") if l.startswith("{")]
base = rows[-1]
for r in rows:
    print(r["lib"], {t: (r[t]["deterministic"], r[t]["per_instance"] == base[t]["per_instance"]) for t in ("full", "masked")}, r["full"]["per_instance"][:2])
PY
timeout 900 python scripts/ab_libs.py build/ab/r06_rec.so build/ab/r06_rec.so:BTBA_NO_PAIR_RECORDS=1 build/ab/r06_rec.so build/ab/r06_rec.so:BTBA_NO_PAIR_RECORDS=1 build/ab/r06_rec.so build/ab/r06_rec.so:BTBA_NO_PAIR_RECORDS=1 > $OUT/ab_records.jsonl 2> $OUT/ab_records.err; python - <<'PY'
import json
for l in open("gpurun_out/r06/ab_records.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r["lib"], r.get("env"), r["full"]["ms_per_step"], r["full"]["sweep_us"], r["full"]["solve_us"], r["full"]["checksum"], "| masked", r["masked"]["ms_per_step"], r["masked"]["sweep_us"], r["masked"]["solve_us"], r["masked"]["checksum"])
    else: print(l[:300])
PY
AB_B=1 timeout 600 python scripts/ab_libs.py build/ab/r06_rec.so build/ab/r06_rec.so:BTBA_NO_PAIR_RECORDS=1 build/ab/r06_rec.so build/ab/r06_rec.so:BTBA_NO_PAIR_RECORDS=1 > $OUT/ab_records_b1.jsonl 2>> $OUT/ab_records.err; python - <<'PY'
import json
for l in open("gpurun_out/r06/ab_records_b1.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print("B=1", r.get("env"), r["full"]["ms_per_step"], r["full"]["sweep_us"], r["full"]["solve_us"], "| masked", r["masked"]["ms_per_step"], r["masked"]["sweep_us"], r["masked"]["solve_us"])
PY
BTBA_LIB_PATH=build/ab/wgtrace.so timeout 200 python scripts/wg_trace.py 2>/dev/null | cut -c1-560
BTBA_NO_PAIR_RECORDS=1 BTBA_LIB_PATH=build/ab/wgtrace.so timeout 200 python scripts/wg_trace.py 2>/dev/null | cut -c1-560
timeout 1800 python -m pytest tests -q -m gpu > $OUT/gputests_8.log 2>&1; tail -8 $OUT/gputests_8.log
