#!/bin/bash
# Round 6, GPU call 21: the taps of a block trip from a per-wave LDS copy of the target patch (-DBTBA_TAP_PATCH=1) against the product, alternating; and how many trips take the patch path.
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
timeout 1500 python scripts/ab_libs.py $B/r6b_base2.so $B/r6b_patch.so $B/r6b_base2.so $B/r6b_patch.so $B/r6b_base2.so $B/r6b_patch.so > $OUT/tap_patch.jsonl 2>&1
cat $OUT/tap_patch.jsonl
BTBA_LIB_PATH=$B/r6b_patchcen.so timeout 300 python scripts/sweep_census.py > $OUT/tap_patch_census.json 2>$OUT/tap_patch_census.err; cat $OUT/tap_patch_census.json; tail -3 $OUT/tap_patch_census.err
