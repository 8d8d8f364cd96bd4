#!/bin/bash
# Round 6, GPU call 15: does a finer-grained tail shorten the fused launch?  sparse chunks per correspondence segment x the share of sparse items closing the launch, c3 x 32.
OUT=gpurun_out/r06; mkdir -p $OUT; rm -f $OUT/c3_chunks.jsonl
timeout 1500 scripts/sweep_matrix.sh $OUT/c3_chunks.jsonl c3 "" "BTBA_BENCH_CHUNKS=2" "BTBA_BENCH_CHUNKS=3" "BTBA_BENCH_CHUNKS=4" "" "BTBA_BENCH_CHUNKS=2" "BTBA_SPARSE_TAIL=192" "BTBA_SPARSE_TAIL=128 BTBA_BENCH_CHUNKS=2" ""
