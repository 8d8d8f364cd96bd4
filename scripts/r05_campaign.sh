#!/bin/bash
# Round 5's measurement campaign on the final build (GPU box): lines of record, rocprofv3 passes for c3 / masked / c2 / c4, workgroup timeline, the solve-kernel A/B.
scripts/r05_lines.sh
timeout 300 scripts/profile_bench.sh r05 > /dev/null 2>&1
timeout 300 scripts/profile_bench.sh r05_masked --masked > /dev/null 2>&1
timeout 300 scripts/profile_bench.sh r05_c2 --config c2 > /dev/null 2>&1
timeout 600 scripts/profile_bench.sh r05_c4 --config c4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
BTBA_LIB_PATH=build/ab/wgtrace.so timeout 200 python scripts/wg_trace.py > gpurun_out/r05_final/wg_trace_c3x32_r05_1tile.json 2>/dev/null
timeout 300 python scripts/ab_solve.py gpurun_out/r05_final/ab_solve.json > gpurun_out/r05_final/ab_solve.log 2>&1
for t in r05 r05_masked r05_c2 r05_c4; do echo "== $t"; head -4 gpurun_out/prof_$t/stats/bench_kernel_stats.csv 2>/dev/null | cut -c1-200; tail -1 gpurun_out/prof_$t/*.log | tail -8 | grep rc=; done
cat gpurun_out/r05_final/wg_trace_c3x32_r05_1tile.json | cut -c1-700
du -sh gpurun_out
