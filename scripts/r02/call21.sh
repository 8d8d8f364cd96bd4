#!/bin/bash
# round 2, GPU call 21: rotated-ray tables + raw v_min in the pixel loop; full GPU suite
mkdir -p gpurun_out/r02_21
O=gpurun_out/r02_21
export AB_NO_TIMING=1
run() { echo "# $*" >> $O/ab.jsonl; env "$@" timeout 200 python scripts/ab_libs.py ${LIB:-build/ab/head3.so} >> $O/ab.jsonl 2>> $O/ab.err; }
LIB=build/ab/head2.so run AB_FLAGS=0
run AB_FLAGS=0
run AB_B=1
unset AB_NO_TIMING
run AB_FLAGS=2048
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
cat $O/ab.jsonl; tail -3 $O/ab.err
