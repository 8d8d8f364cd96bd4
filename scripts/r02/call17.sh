#!/bin/bash
# round 2, GPU call 17: heaviest-pairs-first work order of the dense items
mkdir -p gpurun_out/r02_17
O=gpurun_out/r02_17
export AB_NO_TIMING=1
run() { echo "# $*" >> $O/ab.jsonl; env "$@" timeout 200 python scripts/ab_libs.py ${LIB:-build/ab/head.so} >> $O/ab.jsonl 2>> $O/ab.err; }
run BTBA_NO_DENSE_ORDER=1
run AB_FLAGS=0
run BTBA_BENCH_TILES=1
run BTBA_BENCH_TILES=3
unset AB_NO_TIMING
run AB_FLAGS=2048
AB_B=1 run AB_FLAGS=2048
AB_B=1 run AB_FLAGS=2048 BTBA_NO_DENSE_ORDER=1
cat $O/ab.jsonl; tail -3 $O/ab.err
