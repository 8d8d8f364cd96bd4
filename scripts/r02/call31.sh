#!/bin/bash
# round 2, GPU call 31: single-instance latency -- is the dead-block pre-pass worth it when the chip is mostly empty?  tile counts at B = 1
mkdir -p gpurun_out/r02_31
O=gpurun_out/r02_31
export AB_NO_TIMING=1 AB_B=1
run() { echo "# $*" >> $O/ab.jsonl; env "$@" timeout 200 python scripts/ab_libs.py build/ab/head.so >> $O/ab.jsonl 2>> $O/ab.err; }
run AB_FLAGS=0
run BTBA_NO_BLOCK_SKIP=1
run BTBA_BENCH_TILES=5
run BTBA_BENCH_TILES=8
run BTBA_BENCH_TILES=15
run BTBA_BENCH_TILES=5 BTBA_NO_BLOCK_SKIP=1
AB_B=2 run AB_FLAGS=0
AB_B=4 run AB_FLAGS=0
AB_B=8 run AB_FLAGS=0
cat $O/ab.jsonl | cut -c1-330
