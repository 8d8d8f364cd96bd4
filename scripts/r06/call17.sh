#!/bin/bash
# Round 6, GPU call 17: the pose / intrinsics operands of the pixel loop as SGPR pairs of packed fp32 instructions (-DBTBA_PK_SGPR=1) against the product, alternating.
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
timeout 1500 python scripts/ab_libs.py $B/r6b_base.so $B/r6b_pk.so $B/r6b_base.so $B/r6b_pk.so $B/r6b_base.so $B/r6b_pk.so > $OUT/pk_sgpr.jsonl 2>&1
cat $OUT/pk_sgpr.jsonl
