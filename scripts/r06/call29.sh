#!/bin/bash
# Round 6, GPU call 29: the issue order of a trip's four tap gathers (rows A, B of the 2 x 2 block).  TCP_PENDING_STALL_CYCLES says the L1 spends a quarter to a third of its cycles stalled on
# requests to lines whose miss is still outstanding: with A A B B the second gather waits for row A's lines before row B's misses go out.  A B A B / A B B A / A A B B (pinned) against the product.
OUT=gpurun_out/r06; mkdir -p $OUT
B=build/ab
timeout 1500 python scripts/ab_libs.py $B/r6b_final.so $B/r6b_to1.so $B/r6b_to2.so $B/r6b_to3.so $B/r6b_final.so $B/r6b_to1.so $B/r6b_to2.so $B/r6b_to3.so $B/r6b_final.so > $OUT/tap_order.jsonl 2>&1
cat $OUT/tap_order.jsonl
