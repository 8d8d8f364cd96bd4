#!/bin/bash
# Round 6, GPU call 23: waves per SIMD of the fused sweep re-measured on the final build (round 3: five -3 %, seven -11 %): -DBTBA_FUSED_WAVES=5 / 7 against the product's six.
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
timeout 1500 python scripts/ab_libs.py $B/r6b_final.so $B/r6b_w7.so $B/r6b_w5.so $B/r6b_final.so $B/r6b_w7.so $B/r6b_w5.so $B/r6b_final.so > $OUT/waves_final.jsonl 2>&1
cat $OUT/waves_final.jsonl
