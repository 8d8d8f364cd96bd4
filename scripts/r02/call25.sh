#!/bin/bash
# round 2, GPU call 25: dense work table in the kernel arguments; finer prologue stamps
mkdir -p gpurun_out/r02_25
O=gpurun_out/r02_25
export AB_NO_TIMING=1
run() { echo "# $*" >> $O/ab.jsonl; env "$@" timeout 200 python scripts/ab_libs.py build/ab/head.so >> $O/ab.jsonl 2>> $O/ab.err; }
run BTBA_NO_KERNARG_TABLE=1
run AB_FLAGS=0
run BTBA_NO_KERNARG_TABLE=1
run AB_FLAGS=0
unset AB_NO_TIMING
BTBA_LIB_PATH=build/ab/trace.so timeout 200 python scripts/wg_trace.py > $O/wg_trace.json 2>> $O/ab.err
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cat $O/ab.jsonl | cut -c1-200; python -c "
import json; d=json.load(open('$O/wg_trace.json')); print(json.dumps(d['dense_phases'])); print(d['span_us'], d['mean_running'])"
