#!/bin/bash
# round 2, GPU call 30: btba_zn_aux (block ranges + valid lists as part of the caller's cache): tests, bench lines, rocprofv3 passes
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r02_30
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > "$O/pytest.log"; tail -2 $O/pytest.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --masked --no-cpu-baseline > $O/bench_masked.json 2> $O/bench_masked.err
timeout 300 python bench.py --latency --no-cpu-baseline > $O/bench_latency.json 2> $O/bench_latency.err
timeout 900 bash scripts/profile_bench.sh r02e > $O/profile.log 2>&1
cd /tmp; export BTBA_BENCH_NPROC=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_r02e_masked/stats" -o bench -- python "$REPO/bench.py" --no-cpu-baseline --distinct 8 --masked > $O/masked_stats.log 2>&1
cd "$REPO"; cut -c1-260 $O/bench_default.json; cut -c1-260 $O/bench_masked.json
