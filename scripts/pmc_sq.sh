#!/bin/bash
# SQ occupancy/stall PMC pass for the sweep kernels (own run: --kernel-trace + --pmc only).
# usage: pmc_sq.sh TAG [lib.so ...]   (each lib is profiled with the same bench command)
set -u
TAG=${1:-sq}; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp BTBA_BENCH_NPROC=1
cd /tmp
ARGS="--steps 2 --warmup 1 --distinct 2 --no-cpu-baseline --no-kernel-timing ${BENCH_EXTRA:-}"
LIBS=("$@"); [ ${#LIBS[@]} -eq 0 ] && LIBS=("$REPO/bundletrack_amd/libbtba.so")
for L in "${LIBS[@]}"; do
  N=$(basename "$L" .so)
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    BTBA_LIB_PATH="$L" timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/${N}_p$i" -o bench -- python "$REPO/bench.py" $ARGS > "$OUT/${N}_p$i.log" 2>&1
    echo "pass $i rc=$?" >> "$OUT/${N}_p$i.log"
    find "$OUT/${N}_p$i" -name "*kernel_trace.csv" -delete
  done
done
du -sh "$OUT"
