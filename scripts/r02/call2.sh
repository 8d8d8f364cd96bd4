#!/bin/bash
# round-2 GPU call 2: full instruction-cost calibration (per-SIMD accounting) + PMC unit pinning
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r02_cal
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_calibrate "$REPO/scripts/valu_calibrate.hip" > "$OUT/cal_build.log" 2>&1
timeout 120 /tmp/valu_calibrate > "$OUT/cal_plain.jsonl" 2> "$OUT/cal_plain.err"
i=0
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_COUNT" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/cal_pmc$i" -o cal -- /tmp/valu_calibrate --pmc > "$OUT/cal_pmc$i.jsonl" 2> "$OUT/cal_pmc$i.err"
  echo "cal pmc $i rc=$?" >> "$OUT/cal_pmc$i.err"
  rm -f "$OUT/cal_pmc$i/"*kernel_trace.csv
done
ls -R "$OUT" | head -30
