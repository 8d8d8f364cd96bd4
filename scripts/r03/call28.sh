#!/bin/bash
# round 3, GPU call 28: 16-lane update phase of k_system_solve: A/B (bit identity via checksums) + latency
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_28
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 500 python scripts/ab_libs.py build/ab/r03s_base.so build/ab/r03s.so build/ab/r03s_base.so build/ab/r03s.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
