#!/bin/bash
# Round 6, GPU call 2: A/B of the sweep's instruction diet (LDS pose operands, tap bases, stream saddr, LUT address, list lanes), the GPU suite with k_solve_mid,
# c4 with / without k_solve_mid, c4 FETCH_SIZE at 1 / 2 / 4 tiles, the 120-window fuzz against the reference and its self-spread.
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 600 python scripts/ab_libs.py build/ab/r06_base.so build/ab/r06_pose.so build/ab/r06_pose_tap.so build/ab/r06_nolist.so build/ab/r06_all.so build/ab/r06_base.so build/ab/r06_all.so > $OUT/ab_diet.jsonl 2> $OUT/ab_diet.err; cat $OUT/ab_diet.jsonl | cut -c1-420
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/gputests_2.log 2>&1; tail -5 $OUT/gputests_2.log
export BTBA_BENCH_CACHE=/tmp/c4_inst.npz
timeout 400 python bench.py --config c4 --no-cpu-baseline --no-tracker-call --steps 30 --warmup 5 > $OUT/bench_c4_mid.json 2>/dev/null; cut -c1-300 $OUT/bench_c4_mid.json; python -c "
import json; d=json.loads(open('$OUT/bench_c4_mid.json').read().strip().splitlines()[-1]); print('c4 mid', d['value'], d['ms_per_step'], d.get('kernels_ms_per_step'))"
BTBA_SOLVE_LEGACY=1 timeout 400 python bench.py --config c4 --no-cpu-baseline --no-tracker-call --steps 30 --warmup 5 > $OUT/bench_c4_legacy.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_c4_legacy.json').read().strip().splitlines()[-1]); print('c4 legacy', d['value'], d['ms_per_step'], d.get('kernels_ms_per_step'))"
export TMPDIR=/tmp; REPO=$PWD
for t in 1 2 4; do
  (cd /tmp && BTBA_BENCH_NPROC=1 BTBA_BENCH_TILES=$t timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/$OUT/c4_fetch_t$t -o bench -- python $REPO/bench.py --config c4 --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-tracker-call --no-incl-pack --no-kernel-timing > $REPO/$OUT/c4_fetch_t$t.log 2>&1)
  find $OUT/c4_fetch_t$t -name "*kernel_trace.csv" -delete; find $OUT/c4_fetch_t$t -name "*agent_info.csv" -delete
  python - $OUT/c4_fetch_t$t $t <<'PY'
import csv, glob, sys
tot = n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fused_sweeps" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print("c4 tiles", sys.argv[2], "FETCH_SIZE per launch (raw counter units):", tot / max(n, 1), "launches", n)
PY
done
unset BTBA_BENCH_CACHE
timeout 1500 python tests/tools/fuzz_parity.py 120 > $OUT/fuzz_parity_120.jsonl 2> $OUT/fuzz.err; tail -1 $OUT/fuzz_parity_120.jsonl
