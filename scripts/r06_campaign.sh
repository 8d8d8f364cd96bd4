#!/bin/bash
# Round 6's measurement campaign on the final build (GPU box).  Counter passes FIRST (rocprofv3 --kernel-trace --stats, then separate --pmc passes: FETCH_SIZE, WRITE_SIZE,
# the SQ set) for c3 / c3 masked / c2 / c4, summarised into profiles/r06/ + profiles/sweep_counters.json on the box's copy of the tree; THEN the lines of record, every one
# with its parity leg, so that each line carries parity, traffic and valu_issue of exactly its workload and kernel-source hash.  Everything lands under gpurun_out/r06_final/.
OUT=gpurun_out/r06_final; mkdir -p $OUT/bench_lines $OUT/profiles
export BTBA_PROFILE_ROUND=r06
timeout 400 scripts/profile_bench.sh r06 > /dev/null 2>&1;                      python scripts/summarize_profiles.py r06 r06 2>&1 | tail -1
timeout 400 scripts/profile_bench.sh r06_masked --masked > /dev/null 2>&1;      python scripts/summarize_profiles.py r06_masked r06 --masked --entryj 2>&1 | tail -1
timeout 400 scripts/profile_bench.sh r06_c2 --config c2 > /dev/null 2>&1;       python scripts/summarize_profiles.py r06_c2 r06 --config c2 2>&1 | tail -1
timeout 900 scripts/profile_bench.sh r06_c4 --config c4 > /dev/null 2>&1;       python scripts/summarize_profiles.py r06_c4 r06 --config c4 2>&1 | tail -1
cd $GRAFT_REPO_ROOT
cp profiles/r06/bench_*_r06*_kernel_stats.csv profiles/r06/bench_*_r06*_pmc_fetch_write.csv profiles/r06/bench_*_r06*_sq_counters.csv $OUT/profiles/ 2>/dev/null; cp profiles/sweep_counters.json $OUT/profiles/
for t in r06 r06_masked r06_c2 r06_c4; do echo "== $t"; head -4 gpurun_out/prof_$t/stats/bench_kernel_stats.csv 2>/dev/null | cut -c1-160; grep -h "rc=" gpurun_out/prof_$t/*.log | tr '\n' ' '; echo; done
timeout 300 python bench.py > $OUT/bench_lines/bench_default.json 2> $OUT/bench_lines/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_lines/bench_driver_style.json 2> $OUT/bench_lines/bench_driver_style.err
timeout 300 python bench.py --masked --no-cpu-baseline --parity --no-tracker-call > $OUT/bench_lines/bench_masked.json 2>/dev/null
timeout 300 python bench.py --config c2 --no-cpu-baseline --parity --no-tracker-call > $OUT/bench_lines/bench_c2.json 2>/dev/null
timeout 600 python bench.py --config c4 --no-cpu-baseline --parity --no-tracker-call > $OUT/bench_lines/bench_c4.json 2>/dev/null
timeout 300 python bench.py --masked --no-cpu-baseline --parity --no-tracker-call --instances 1 --distinct 1 > $OUT/bench_lines/bench_masked_single.json 2>/dev/null
BTBA_LIB_PATH=build/ab/wgtrace.so timeout 200 python scripts/wg_trace.py > $OUT/wg_trace_c3x32_r06_1tile.json 2>/dev/null
timeout 400 python scripts/ab_solve.py $OUT/ab_solve.json > $OUT/ab_solve.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_final/bench_lines/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("value_incl_pack"), r.get("frac"), (r.get("executed") or {}).get("frac"), r.get("traffic"), (r.get("valu_issue") or {}).get("busy_frac"),
              (r.get("hbm_algorithmic") or {}).get("traffic_over_compulsory"), (d.get("kernels_ms_per_step") or {}).get("system_solve"), d.get("single_instance", {}).get("ms_per_solve"), (d.get("parity") or {}).get("ok"), (d.get("parity") or {}).get("worst_rot"))
    except Exception as e:
        print(f, "FAILED", e)
PY
cut -c1-600 $OUT/wg_trace_c3x32_r06_1tile.json; du -sh gpurun_out
# the parity records: the per-iterate and random-window tests with their printed figures, the 120-window fuzz against the reference and its self-spread
timeout 900 python -m pytest tests/test_gpu_vs_reference.py -q -m gpu -s -k "every_gauss_newton or random_windows or matches_the_reference_solver" 2>&1 | grep -v "^$" | grep -i "vs the reference\|random windows\|leave the bar\|within 3x\|beyond\|passed\|failed" > $OUT/reference_tests_printed.txt; cat $OUT/reference_tests_printed.txt | cut -c1-400
timeout 1800 python tests/tools/fuzz_parity.py 120 > $OUT/fuzz_parity_120.jsonl 2> $OUT/fuzz.err; tail -1 $OUT/fuzz_parity_120.jsonl
timeout 600 python scripts/boundary_timing.py > $OUT/boundary_timing.jsonl 2>/dev/null; python - <<'PY'
import json
for l in open("gpurun_out/r06_final/boundary_timing.jsonl"):
    r = json.loads(l); print(r["K"], r["corr_per_pair"], r["valid_fraction"], "stateless", r["wall_ms_median"], "keyed", r["wall_ms_median_keyed"], "keyed+corr", r["wall_ms_median_keyed_frames_and_correspondences"])
PY
timeout 1800 python -m pytest tests -q -m gpu > $OUT/gputests.log 2>&1; tail -3 $OUT/gputests.log
