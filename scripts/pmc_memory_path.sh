#!/bin/bash
# Memory-path and LDS counters of the bench workload's kernels (rounds 4-5) -- each set its own rocprofv3 run (--kernel-trace + --pmc only).
# usage: scripts/pmc_memory_path.sh <tag> [bench args]      then: python scripts/summarize_pmc.py gpurun_out/pmc_<tag> profiles/r05/memory_path_counters_<tag>.json
set -u
TAG=${1:-mem}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
export BTBA_BENCH_CACHE=/tmp/bench_instances_pmc_$TAG.npz
BTBA_BENCH_NPROC=8 timeout 300 python "$REPO/bench.py" --no-cpu-baseline --steps 2 --warmup 1 --settle-ms 0 $* > "$OUT/pre.log" 2>&1
export BTBA_BENCH_NPROC=1
ARGS="--steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-tracker-call --no-incl-pack --no-kernel-timing $*"
echo "bench args: $ARGS" > "$OUT/args.txt"
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  echo "$SET" > "$OUT/p$i.set"
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -o bench -- python "$REPO/bench.py" $ARGS > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$?" >> "$OUT/p$i.log"
  find "$OUT/p$i" -name "*kernel_trace.csv" -delete
  find "$OUT/p$i" -name "*agent_info.csv" -delete
done
rm -f "$BTBA_BENCH_CACHE"
du -sh "$OUT"
