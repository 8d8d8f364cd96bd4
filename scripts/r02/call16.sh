#!/bin/bash
# round 2, GPU call 16: persistent workgroups pulling sweep items from per-XCD queues vs the plain launch
mkdir -p gpurun_out/r02_16
O=gpurun_out/r02_16
export AB_NO_TIMING=1
run() { echo "# $*" >> $O/ab.jsonl; env "$@" timeout 200 python scripts/ab_libs.py ${LIB:-build/ab/persist.so} >> $O/ab.jsonl 2>> $O/ab.err; }
LIB=build/ab/head.so run AB_FLAGS=0
run BTBA_PERSIST_WGS=1536
run BTBA_PERSIST_WGS=2048
run BTBA_PERSIST_WGS=1536 BTBA_BENCH_TILES=3
run BTBA_PERSIST_WGS=1536 BTBA_BENCH_TILES=4
unset AB_NO_TIMING
LIB=build/ab/head.so run AB_FLAGS=2048
cat $O/ab.jsonl; tail -3 $O/ab.err
