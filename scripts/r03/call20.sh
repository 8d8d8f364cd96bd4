#!/bin/bash
# round 3, GPU call 20: BTBA_REDUCE_ATOMIC: test + timing against the deterministic mode
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_20
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "atomic or batch_equals or tile_and_chunk" -s 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; grep -i "atomic vs\|passed\|failed\|Error" $O/pytest.log | head -20
timeout 500 python scripts/ab_libs.py build/ab/r03q.so build/ab/r03q.so:AB_REDUCTION=1 build/ab/r03q.so build/ab/r03q.so:AB_REDUCTION=1 > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
