#!/bin/bash
# round-2 GPU call 3: A/B of the re-shaped dense sweep, calibrated VALU-busy counters, parity tests on the new kernel
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r02_ab1
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
timeout 600 python scripts/ab_libs.py build/ab/base.so build/ab/A.so build/ab/B.so build/ab/C.so build/ab/D.so > "$OUT/ab.jsonl" 2> "$OUT/ab.err"
cat "$OUT/ab.jsonl"
cd /tmp
export BTBA_BENCH_NPROC=1
for L in base A; do
  BTBA_LIB_PATH=$REPO/build/ab/$L.so timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_$L" -o bench -- python "$REPO/bench.py" --steps 2 --warmup 1 --distinct 2 --no-cpu-baseline --no-kernel-timing > "$OUT/pmc_$L.log" 2>&1
  find "$OUT/pmc_$L" -name "*kernel_trace.csv" -delete
done
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_golden.py tests/test_gpu_fullsize.py tests/test_gpu_vs_reference.py -m gpu -q -x 2>&1 | tail -25 > "$OUT/pytest.log"
cat "$OUT/pytest.log"
