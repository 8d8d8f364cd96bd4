#!/bin/bash
# Round 6, GPU call 19: taps only for the lanes still alive (-DBTBA_TAPS_EXEC=1) against the product, alternating.
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
timeout 1500 python scripts/ab_libs.py $B/r6b_base.so $B/r6b_tex.so $B/r6b_base.so $B/r6b_tex.so $B/r6b_base.so $B/r6b_tex.so > $OUT/taps_exec.jsonl 2>&1
cat $OUT/taps_exec.jsonl
