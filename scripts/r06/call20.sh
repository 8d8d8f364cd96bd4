#!/bin/bash
# Round 6, GPU call 20: LDS sensitivity of the pixel loop -- 4 / 8 more conflict-free 16-byte-per-lane LDS reads per trip (+32 / +64 LDS cycles per trip and CU), same results, against the product.
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
timeout 1500 python scripts/ab_libs.py $B/r6b_base2.so $B/r6b_xl4.so $B/r6b_xl8.so $B/r6b_base2.so $B/r6b_xl4.so $B/r6b_xl8.so $B/r6b_base2.so > $OUT/lds_probe.jsonl 2>&1
cat $OUT/lds_probe.jsonl
