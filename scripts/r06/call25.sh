#!/bin/bash
# Round 6, GPU call 25: do the waves of a workgroup walk their blocks in a convoy (all gathers of a compute unit queued at once, then all of the arithmetic)?  wave w of a workgroup starts
# its walk 256 w / 640 w cycles late (-DBTBA_STAGGER=4 / 10), against the product.
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
timeout 1500 python scripts/ab_libs.py $B/r6b_final.so $B/r6b_st4.so $B/r6b_st10.so $B/r6b_final.so $B/r6b_st4.so $B/r6b_st10.so $B/r6b_final.so > $OUT/stagger.jsonl 2>&1
cat $OUT/stagger.jsonl
