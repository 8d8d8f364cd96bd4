#!/bin/bash
# Extra PMC passes for the dense sweep (each pass is its own run: --kernel-trace + --pmc only).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp BTBA_BENCH_NPROC=1
cd /tmp
timeout 60 rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
ARGS="--steps 2 --warmup 1 --distinct 2 --no-cpu-baseline --no-kernel-timing"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -o bench -- python "$REPO/bench.py" $ARGS > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$?" >> "$OUT/p$i.log"
  find "$OUT/p$i" -name "*kernel_trace.csv" -delete
done
du -sh "$OUT"; ls -R "$OUT" | head -40
