#!/bin/bash
# Round 6, GPU call 18: which pipe does the pixel loop wait for?  Sensitivity probes with UNCHANGED results: 2 / 4 more 16-byte gathers per trip (LDS-direct into a sink, no VGPRs:
# texture addresser + 40 % / + 80 %), 21 more scalar-operand FMAs per trip (vector issue + ~ 20 %), against the product, alternating.
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
timeout 1500 python scripts/ab_libs.py $B/r6b_base.so $B/r6b_xt2.so $B/r6b_xt4.so $B/r6b_xf.so $B/r6b_base.so $B/r6b_xt2.so $B/r6b_xt4.so $B/r6b_xf.so $B/r6b_base.so > $OUT/bound_probes.jsonl 2>&1
cat $OUT/bound_probes.jsonl
