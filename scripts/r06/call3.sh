#!/bin/bash
# Round 6, GPU call 3: which of the sweep-diet switches are deterministic / bit-identical / faster, one at a time; pose operands in VGPRs at five waves per SIMD.
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python scripts/r06/determinism.py build/ab/r06_base.so build/ab/r06_pose.so build/ab/r06_tap.so build/ab/r06_saddr.so build/ab/r06_lut.so build/ab/r06_list.so build/ab/r06_all.so build/ab/r06_vgpr5.so > $OUT/determinism.jsonl 2> $OUT/determinism.err; python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r06/determinism.jsonl") if l.startswith("{")]
base = rows[0]
for r in rows:
    print(r["lib"], {t: (r[t]["deterministic"], r[t]["per_instance"] == base[t]["per_instance"]) for t in ("full", "masked")}, r["full"]["per_instance"][:2], r["full"]["run2"][:2] if r["full"]["run2"] else None)
PY
timeout 900 python scripts/ab_libs.py build/ab/r06_base.so build/ab/r06_tap.so build/ab/r06_saddr.so build/ab/r06_lut.so build/ab/r06_list.so build/ab/r06_base5.so build/ab/r06_vgpr5.so build/ab/r06_base.so build/ab/r06_vgpr5.so > $OUT/ab_diet2.jsonl 2> $OUT/ab_diet2.err; python - <<'PY'
import json
for l in open("gpurun_out/r06/ab_diet2.jsonl"):
    if l.startswith("{"):
        r = json.loads(l); print(r["lib"], r["full"]["ms_per_step"], r["full"]["sweep_us"], r["full"]["checksum"], "| masked", r["masked"]["ms_per_step"], r["masked"]["sweep_us"], r["masked"]["checksum"])
    else: print(l[:300])
PY
