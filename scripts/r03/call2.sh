#!/bin/bash
# round 3, GPU call 2: persistent fused sweep (k_fused_persist) A/B: v21 | descriptors, one workgroup per item | descriptors, persistent
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_02
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 500 python scripts/ab_libs.py build/ab/v21.so build/ab/r03b.so:BTBA_NO_PERSISTENT=1 build/ab/r03b.so build/ab/v21.so build/ab/r03b.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -v "^$" > "$O/pytest.log"; echo "pytest rc=${PIPESTATUS[0]}" | tee -a $O/pytest.log; tail -5 $O/pytest.log
