#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r02_ab5
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
cat /sys/fs/cgroup/cpu.max > "$OUT/cpu_max.txt" 2>&1; nproc >> "$OUT/cpu_max.txt"
timeout 300 python scripts/ab_libs.py build/ab/M.so build/ab/Q.so > "$OUT/ab.jsonl" 2> "$OUT/ab.err"; cat "$OUT/ab.jsonl"
timeout 900 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_multirank.py tests/test_tracking_session.py tests/test_gpu_ransac.py tests/test_depth_processing.py tests/test_cpp_host.py tests/test_reference_golden.py tests/test_problem_dump.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -40 > "$OUT/pytest.log"
tail -30 "$OUT/pytest.log"
timeout 200 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -4 "$OUT/bench.err"
