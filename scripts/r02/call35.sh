#!/bin/bash
# round 2, GPU call 35: final state of the round -- full GPU suite, bench lines (c3 default / masked / latency, c2, c4), rocprofv3 passes,
# boundary and large-window timings
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r02_35
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > "$O/pytest_full.log"; tail -2 $O/pytest_full.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --masked --no-cpu-baseline > $O/bench_masked.json 2> $O/bench_masked.err
timeout 300 python bench.py --latency --no-cpu-baseline > $O/bench_latency.json 2> $O/bench_latency.err
timeout 300 python bench.py --config c2 --no-cpu-baseline --distinct 8 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 400 python bench.py --config c4 --no-cpu-baseline --distinct 4 --steps 10 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 900 bash scripts/profile_bench.sh r02g > $O/profile.log 2>&1
cd /tmp; export BTBA_BENCH_NPROC=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_r02g_masked/stats" -o bench -- python "$REPO/bench.py" --no-cpu-baseline --distinct 8 --masked > $O/masked_stats.log 2>&1
cd "$REPO"
timeout 300 python scripts/boundary_timing.py > $O/boundary_timing.jsonl 2> $O/boundary.err
timeout 300 python scripts/large_window_timing.py > $O/large_window_timing.jsonl 2> $O/lw.err
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
