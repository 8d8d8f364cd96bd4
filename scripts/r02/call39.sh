#!/bin/bash
# round 2, GPU call 39: k_system_solve reads the sparse and dense sweep partials in ONE round of loads (A/B against the previous build
# in build/libbtba_base.so: batch and single-instance bench lines), then the full GPU suite on the new build
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r02_39
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
for v in base new base new; do
  if [ $v = base ]; then export BTBA_LIB_PATH=$REPO/build/libbtba_base.so; else unset BTBA_LIB_PATH; fi
  timeout 200 python bench.py --no-cpu-baseline --latency >> $O/bench_$v.jsonl 2>> $O/bench_$v.err
done
unset BTBA_LIB_PATH
python - <<'P'
import json,glob
for v in ("base","new"):
    for l in open(f"gpurun_out/r02_39/bench_{v}.jsonl"):
        d=json.loads(l); ks=d.get("kernels",{}) or {}
        print(v, d["value"], d["ms_per_step"], {k:d[k] for k in d if "latency" in k}, d["roofline"].get("kernel_us"), [ (k,v2) for k,v2 in d.items() if k.startswith("ms_") ][:6])
P
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^$" > "$O/pytest_full.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest_full.log; tail -3 $O/pytest_full.log
