#!/bin/bash
# round 2, GPU call 14: folded wave reduction + parallel block ranges vs call 13's build; workgroup timeline of the fused sweep
mkdir -p gpurun_out/r02_14
O=gpurun_out/r02_14
timeout 300 python scripts/ab_libs.py build/ab/hull.so build/ab/head.so > $O/ab.jsonl 2> $O/ab.err
AB_B=1 timeout 200 python scripts/ab_libs.py build/ab/hull.so build/ab/head.so > $O/ab_b1.jsonl 2>> $O/ab.err
for t in 0 3 4; do BTBA_LIB_PATH=build/ab/trace.so timeout 200 python scripts/wg_trace.py --tiles $t >> $O/wg_trace.jsonl 2>> $O/ab.err; done
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
cat $O/ab.jsonl $O/ab_b1.jsonl $O/wg_trace.jsonl; tail -5 $O/ab.err
