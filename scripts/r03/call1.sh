#!/bin/bash
# round 3, GPU call 1: k_pair_setup + descriptor-driven dense workgroups -- full GPU test suite, A/B against v21 on one box, default bench line
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_01
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; tail -15 $O/pytest.log
timeout 400 python scripts/ab_libs.py build/ab/v21.so build/ab/r03a.so build/ab/v21.so build/ab/r03a.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"
timeout 300 python bench.py --no-cpu-baseline --latency > "$O/bench.json" 2> "$O/bench.err"; tail -c 1500 "$O/bench.json"
