#!/bin/bash
# round 2, GPU call 20: phase stamps inside the dense workgroups (prologue / pixel loop / epilogue)
mkdir -p gpurun_out/r02_20
O=gpurun_out/r02_20
for t in 0 3; do BTBA_LIB_PATH=build/ab/trace.so timeout 200 python scripts/wg_trace.py --tiles $t >> $O/wg_trace.jsonl 2>> $O/err.log; done
cat $O/wg_trace.jsonl; tail -3 $O/err.log
