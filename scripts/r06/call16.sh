#!/bin/bash
# Round 6, GPU call 16: what would an L2 that held every tap be worth?  A batch of 32 IDENTICAL c3 instances, once reading 32 copies of the frames (the product) and once
# reading ONE copy (-DBTBA_DEV_ALIAS=1: same bits, frame working set 4.6 MB on the whole chip); and the source block as a non-temporal load.
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
timeout 1500 python scripts/ab_libs.py $B/r6b_base.so:AB_DISTINCT=1 $B/r6b_alias1.so:AB_DISTINCT=1 $B/r6b_base.so $B/r6b_srcnt.so \
    $B/r6b_base.so:AB_DISTINCT=1 $B/r6b_alias1.so:AB_DISTINCT=1 $B/r6b_srcnt.so $B/r6b_base.so > $OUT/l2_alias_probe.jsonl 2>&1
cat $OUT/l2_alias_probe.jsonl
