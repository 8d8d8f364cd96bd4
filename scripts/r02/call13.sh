#!/bin/bash
# round 2, GPU call 13: dead-block skip (hull test) in the dense block walk -- A/B against the previous build, exactness test, parity
mkdir -p gpurun_out/r02_13
O=gpurun_out/r02_13
timeout 300 python scripts/ab_libs.py build/ab/base.so build/ab/hull.so > $O/ab.jsonl 2> $O/ab.err
for t in 3 8 15; do BTBA_BENCH_TILES=$t timeout 200 python scripts/ab_libs.py build/ab/hull.so >> $O/ab_tiles.jsonl 2>> $O/ab.err; done
AB_B=1 timeout 200 python scripts/ab_libs.py build/ab/base.so build/ab/hull.so > $O/ab_b1.jsonl 2>> $O/ab.err
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -s > $O/pytest.log 2>&1
tail -5 $O/pytest.log
cat $O/ab.jsonl $O/ab_tiles.jsonl $O/ab_b1.jsonl
