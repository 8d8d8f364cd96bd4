#!/bin/bash
# round 2, GPU call 32: tile counts normalised to block rows (latency mode), k_system_solve phase stamps on large windows
mkdir -p gpurun_out/r02_32
O=gpurun_out/r02_32
timeout 300 python scripts/sys_clocks_big.py > $O/sys_clocks_big.txt 2>&1; tail -4 $O/sys_clocks_big.txt
export AB_NO_TIMING=1
run() { echo "# $*" >> $O/ab.jsonl; env "$@" timeout 200 python scripts/ab_libs.py build/ab/head.so >> $O/ab.jsonl 2>> $O/ab.err; }
AB_B=1 run AB_FLAGS=0
AB_B=2 run AB_FLAGS=0
AB_B=4 run AB_FLAGS=0
run AB_FLAGS=0
cat $O/ab.jsonl | cut -c1-330
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
