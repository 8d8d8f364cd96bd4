#!/bin/bash
# round 2, GPU call 24: final numbers of the round -- bench lines, rocprofv3 passes of the bench command, boundary / workgroup timelines, full GPU suite
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r02_24
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --masked --no-cpu-baseline > $O/bench_masked.json 2> $O/bench_masked.err
timeout 300 python bench.py --latency --no-cpu-baseline > $O/bench_latency.json 2> $O/bench_latency.err
timeout 900 bash scripts/profile_bench.sh r02c > $O/profile.log 2>&1
cd "$REPO"
timeout 300 python scripts/boundary_timing.py > $O/boundary_timing.jsonl 2> $O/boundary.err
BTBA_LIB_PATH=build/ab/trace.so timeout 200 python scripts/wg_trace.py > $O/wg_trace.json 2>> $O/wg.err
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > "$O/pytest_full.log"
tail -2 $O/pytest_full.log
cut -c1-300 $O/bench_default.json; cut -c1-600 $O/boundary_timing.jsonl
