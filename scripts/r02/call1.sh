#!/bin/bash
# round-2 GPU call 1: counter calibration + SQ passes of the current kernels + a reference bench line
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r02_call1
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_calibrate "$REPO/scripts/valu_calibrate.hip" > "$OUT/cal_build.log" 2>&1
/tmp/valu_calibrate > "$OUT/cal_plain.jsonl" 2> "$OUT/cal_plain.err"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
i=0
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/cal_pmc$i" -o cal -- /tmp/valu_calibrate --pmc > "$OUT/cal_pmc$i.jsonl" 2> "$OUT/cal_pmc$i.err"
  echo "cal pmc $i rc=$?" >> "$OUT/cal_pmc$i.err"
done
cd "$REPO"
python bench.py --no-cpu-baseline > "$OUT/bench_base.json" 2> "$OUT/bench_base.err"
BENCH_EXTRA="" bash scripts/pmc_sq.sh r02base > "$OUT/pmc_sq.log" 2>&1
ls -R "$OUT" | head -50
tail -3 "$OUT/bench_base.json" | cut -c1-600
