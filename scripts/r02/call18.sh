#!/bin/bash
# round 2, GPU call 18: work table (one load), M staged at workgroup start; workgroup timeline again
mkdir -p gpurun_out/r02_18
O=gpurun_out/r02_18
export AB_NO_TIMING=1
run() { echo "# $*" >> $O/ab.jsonl; env "$@" timeout 200 python scripts/ab_libs.py ${LIB:-build/ab/head2.so} >> $O/ab.jsonl 2>> $O/ab.err; }
LIB=build/ab/head.so run AB_FLAGS=0
run AB_FLAGS=0
run BTBA_BENCH_TILES=3
unset AB_NO_TIMING
BTBA_LIB_PATH=build/ab/trace.so timeout 200 python scripts/wg_trace.py > $O/wg_trace.jsonl 2>> $O/ab.err
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
cat $O/ab.jsonl $O/wg_trace.jsonl; tail -3 $O/ab.err
