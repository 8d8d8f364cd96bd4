#!/bin/bash
# round 2, GPU call 28: PCG on four waves meeting on LDS flags
mkdir -p gpurun_out/r02_28
O=gpurun_out/r02_28
run() { echo "# $*" >> $O/ab.jsonl; env "$@" timeout 200 python scripts/ab_libs.py $LIB >> $O/ab.jsonl 2>> $O/ab.err; }
LIB=build/ab/base.so run AB_FLAGS=2048
LIB=build/ab/head.so run AB_FLAGS=2048
LIB=build/ab/head2.so run AB_FLAGS=2048
LIB=build/ab/base.so run AB_FLAGS=2048 AB_B=1
LIB=build/ab/head2.so run AB_FLAGS=2048 AB_B=1
timeout 200 python scripts/sys_clocks.py > $O/sys_clocks.txt 2>&1; tail -2 $O/sys_clocks.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_vs_reference.py tests/test_tracking_session.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cat $O/ab.jsonl | cut -c1-330
