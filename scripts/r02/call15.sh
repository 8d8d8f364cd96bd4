#!/bin/bash
# round 2, GPU call 15: instance groups on several streams (sweep tails / k_system_solve of one group under another group's sweep)
mkdir -p gpurun_out/r02_15
O=gpurun_out/r02_15
export AB_NO_TIMING=1
run() { echo "# $*" >> $O/ab.jsonl; env "$@" timeout 200 python scripts/ab_libs.py build/ab/head.so >> $O/ab.jsonl 2>> $O/ab.err; }
run AB_FLAGS=0
run AB_FLAGS=32 BTBA_GROUPS=2
run AB_FLAGS=32 BTBA_GROUPS=2 BTBA_GROUP_PRIO=equal
run AB_FLAGS=32 BTBA_GROUPS=4
run AB_FLAGS=32 BTBA_GROUPS=4 BTBA_GROUP_PRIO=equal
run AB_FLAGS=32 BTBA_GROUPS=8 BTBA_GROUP_PRIO=equal
cat $O/ab.jsonl; tail -3 $O/ab.err
