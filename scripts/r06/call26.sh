#!/bin/bash
# Round 6, GPU call 26: the alias probe at c4 x 32 (K = 30: an instance's frames are 9.2 MB against a 4 MB L2, the batch's 295 MB against the 256 MB memory-side cache; L2 hit rate 0.61):
# 32 identical instances reading 32 copies of the frames or ONE (-DBTBA_DEV_ALIAS=1, same bits).  Builds of the first session's product (no exec mask on the taps).
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
AB_CONFIG=c4 timeout 2400 python scripts/ab_libs.py $B/r6b_base.so:AB_DISTINCT=1 $B/r6b_alias1.so:AB_DISTINCT=1 $B/r6b_base.so:AB_DISTINCT=1 $B/r6b_alias1.so:AB_DISTINCT=1 $B/r6b_base.so > $OUT/l2_alias_probe_c4.jsonl 2>&1
cat $OUT/l2_alias_probe_c4.jsonl
