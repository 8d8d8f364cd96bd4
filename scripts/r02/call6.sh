#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r02_ab4
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
timeout 600 python scripts/ab_libs.py build/ab/J.so build/ab/M.so build/ab/N.so > "$OUT/ab.jsonl" 2> "$OUT/ab.err"
BTBA_NO_BLOCK_WALK=1 timeout 200 python scripts/ab_libs.py build/ab/M.so >> "$OUT/ab.jsonl" 2>> "$OUT/ab.err"
for t in 1 3 4; do BTBA_BENCH_TILES=$t timeout 200 python scripts/ab_libs.py build/ab/M.so >> "$OUT/ab_tiles.jsonl" 2>> "$OUT/ab.err"; done
cat "$OUT/ab.jsonl" "$OUT/ab_tiles.jsonl"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_golden.py tests/test_gpu_fullsize.py tests/test_gpu_vs_reference.py -m gpu -q -x 2>&1 | tail -25 > "$OUT/pytest.log"
cat "$OUT/pytest.log"
