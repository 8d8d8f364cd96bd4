#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/r02_ab2
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
timeout 600 python scripts/ab_libs.py build/ab/base.so build/ab/C.so build/ab/E.so build/ab/F.so build/ab/G.so > "$OUT/ab.jsonl" 2> "$OUT/ab.err"
for t in 1 3 4; do BTBA_BENCH_TILES=$t timeout 200 python scripts/ab_libs.py build/ab/F.so >> "$OUT/ab_tiles.jsonl" 2>> "$OUT/ab.err"; done
cat "$OUT/ab.jsonl" "$OUT/ab_tiles.jsonl"
cd /tmp
export BTBA_BENCH_NPROC=1
for L in F; do
  BTBA_LIB_PATH=$REPO/build/ab/$L.so timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_$L" -o bench -- python "$REPO/bench.py" --steps 2 --warmup 1 --distinct 2 --no-cpu-baseline --no-kernel-timing > "$OUT/pmc_$L.log" 2>&1
  find "$OUT/pmc_$L" -name "*kernel_trace.csv" -delete
done
