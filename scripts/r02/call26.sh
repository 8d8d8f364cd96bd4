#!/bin/bash
# round 2, GPU call 26: prologue loads issued first + block ranges prefetched
mkdir -p gpurun_out/r02_26
O=gpurun_out/r02_26
export AB_NO_TIMING=1
run() { echo "# $*" >> $O/ab.jsonl; env "$@" timeout 200 python scripts/ab_libs.py $LIB >> $O/ab.jsonl 2>> $O/ab.err; }
LIB=build/ab/head.so run AB_FLAGS=0
LIB=build/ab/head2.so run AB_FLAGS=0
LIB=build/ab/head.so run AB_FLAGS=0
LIB=build/ab/head2.so run AB_FLAGS=0
LIB=build/ab/head2.so run AB_B=1
LIB=build/ab/head.so run AB_B=1
unset AB_NO_TIMING
BTBA_LIB_PATH=build/ab/trace.so timeout 200 python scripts/wg_trace.py > $O/wg_trace.json 2>> $O/ab.err
cat $O/ab.jsonl | cut -c1-330; python -c "
import json; d=json.load(open('$O/wg_trace.json')); print(json.dumps(d['dense_phases'])); print(d['span_us'], d['mean_running'])"
