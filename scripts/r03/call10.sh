#!/bin/bash
# round 3, GPU call 10: tile count and s_setprio experiments on the fused sweep
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_10
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 800 python scripts/ab_libs.py build/ab/r03g.so build/ab/r03h.so build/ab/r03g.so:BTBA_BENCH_TILES=1 build/ab/r03g.so:BTBA_BENCH_TILES=3 build/ab/r03g.so:BTBA_BENCH_TILES=4 build/ab/r03h.so:BTBA_BENCH_TILES=3 build/ab/r03g.so build/ab/r03h.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
