#!/bin/bash
# Round 6, GPU call 1: the GPU suite on the round's first changes, a driver-style line, the lane census (c3, c4), the c4 tile sweep.
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu -x > $OUT/gputests_1.log 2>&1; tail -5 $OUT/gputests_1.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style_1.json 2> $OUT/bench_driver_style_1.err; cut -c1-400 $OUT/bench_driver_style_1.json
BTBA_LIB_PATH=build/ab/census.so timeout 300 python scripts/sweep_census.py > $OUT/sweep_census_c3.json 2> $OUT/census.err; cat $OUT/sweep_census_c3.json
BTBA_LIB_PATH=build/ab/census.so timeout 400 python scripts/sweep_census.py --config c4 > $OUT/sweep_census_c4.json 2>> $OUT/census.err; cat $OUT/sweep_census_c4.json
timeout 900 scripts/sweep_matrix.sh $OUT/c4_tiles.jsonl c4 "" "BTBA_BENCH_TILES=1" "BTBA_BENCH_TILES=2" "BTBA_BENCH_TILES=3" "BTBA_BENCH_TILES=4" "BTBA_BENCH_TILES=5" \
    "BTBA_PAIR_MAJOR=1 BTBA_BENCH_TILES=2" "BTBA_PAIR_MAJOR=1 BTBA_BENCH_TILES=3" "BTBA_PAIR_MAJOR=1 BTBA_BENCH_TILES=4" "BTBA_PAIR_MAJOR=1 BTBA_BENCH_TILES=5"
