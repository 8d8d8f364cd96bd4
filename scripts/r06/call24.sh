#!/bin/bash
# Round 6, GPU call 24: the texture addresser's and the L1's own counters for the bound probes -- the product of the first session (build/ab/r6b_base.so), the same with two more gathers per
# trip (r6b_xt2.so) and the final build, c3 x 32, one rocprofv3 --pmc pass each (TA_BUSY, TCP stalls; kernel trace only beside it).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06/ta_counters; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
export BTBA_BENCH_CACHE=/tmp/bench_instances_ta.npz
BTBA_BENCH_NPROC=8 timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --settle-ms 0 > $OUT/pre.log 2>&1
export BTBA_BENCH_NPROC=1
ARGS="--steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-tracker-call --no-incl-pack --no-kernel-timing --no-single-instance"
for V in base xt2 final; do
  mkdir -p $OUT/$V; echo "bench args: $ARGS; library build/ab/r6b_$V.so" > $OUT/$V/args.txt
  BTBA_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/r6b_$V.so timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $OUT/$V/p1 -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/$V/p1.log 2>&1
  BTBA_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/r6b_$V.so timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_max TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum --output-format csv -d $OUT/$V/p2 -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/$V/p2.log 2>&1
  find $OUT/$V -name "*kernel_trace.csv" -delete; find $OUT/$V -name "*agent_info.csv" -delete
  python $GRAFT_REPO_ROOT/scripts/summarize_pmc.py $OUT/$V $OUT/$V.json | head -30
  tail -2 $OUT/$V/p2.log
done
rm -f $BTBA_BENCH_CACHE; du -sh $OUT
