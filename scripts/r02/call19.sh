#!/bin/bash
# round 2, GPU call 19: where the kernel arguments live (HIP_FORCE_DEV_KERNARG) vs workgroup start-up latency
mkdir -p gpurun_out/r02_19
O=gpurun_out/r02_19
export AB_NO_TIMING=1
run() { echo "# $*" >> $O/ab.jsonl; env "$@" timeout 200 python scripts/ab_libs.py ${LIB:-build/ab/head2.so} >> $O/ab.jsonl 2>> $O/ab.err; }
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0 AB_B=1
run HIP_FORCE_DEV_KERNARG=1 AB_B=1
run AB_FLAGS=0
cat $O/ab.jsonl; tail -3 $O/ab.err
