#!/bin/bash
# round 2, GPU call 36 (the round's last ~14 GPU-minutes): the RANSAC's default sample stream is now the reference's cuRAND XORWOW
# table, DeviceGuard in every entry point, C++ Bundler session -- full GPU suite first, then the rocprofv3 passes for the new
# source hash (profiles/sweep_counters.json is keyed by it), the bench line, RANSAC timing, smoke.
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r02_36
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
date +%s > $O/t0
timeout 420 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^$" > "$O/pytest_full.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest_full.log; tail -3 $O/pytest_full.log
date +%s > $O/t1
timeout 420 bash scripts/profile_bench.sh r02h > $O/profile.log 2>&1; echo "profile rc=$?"
date +%s > $O/t2
timeout 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.json
timeout 120 python tests/tools/ransac_timing.py > $O/ransac_timing.jsonl 2> $O/ransac.err; echo "ransac rc=$?"
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
date +%s > $O/t3
