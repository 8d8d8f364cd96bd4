#!/bin/bash
# Round 6, GPU call 22: where does the LDS tap patch lose?  2: bounding box only; 3: box + patch loads + wait, taps gathered all the same; 1: the full path.
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
timeout 1500 python scripts/ab_libs.py $B/r6b_base2.so $B/r6b_patch2.so $B/r6b_patch3.so $B/r6b_patch.so $B/r6b_base2.so $B/r6b_patch2.so $B/r6b_patch3.so $B/r6b_patch.so > $OUT/tap_patch_parts.jsonl 2>&1
cat $OUT/tap_patch_parts.jsonl
