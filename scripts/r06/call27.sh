#!/bin/bash
# Round 6, GPU call 27: the four waves of a workgroup walk adjacent blocks -- kept in step by a workgroup barrier every trip / every fourth trip (-DBTBA_LOCKSTEP=1 / 4) their tap patches
# share cache lines (5 lines per patch row instead of 8).  Same bits.  Against the product.
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
timeout 1500 python scripts/ab_libs.py $B/r6b_final.so $B/r6b_ls1.so $B/r6b_ls4.so $B/r6b_final.so $B/r6b_ls1.so $B/r6b_ls4.so $B/r6b_final.so > $OUT/lockstep.jsonl 2>&1
cat $OUT/lockstep.jsonl
