#!/bin/bash
# round 2, GPU call 22: bench lines (default, masked, latency), rocprofv3 passes of the bench command, test log with prints, smoke
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r02_22
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --masked --no-cpu-baseline > $O/bench_masked.json 2> $O/bench_masked.err
timeout 300 python bench.py --latency --no-cpu-baseline > $O/bench_latency.json 2> $O/bench_latency.err
timeout 900 bash scripts/profile_bench.sh r02b > $O/profile.log 2>&1
cd "$REPO"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > "$O/pytest_full.log"
tail -2 $O/pytest_full.log
cut -c1-1500 $O/bench_default.json
