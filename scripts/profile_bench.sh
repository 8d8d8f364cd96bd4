#!/bin/bash
# Profiles the bench workload on the GPU box with rocprofv3.  Writes under gpurun_out/prof_<tag>/ :
#   stats/   --kernel-trace --stats            (per-kernel durations, bench.py's default step counts)
#   fetch/   --kernel-trace --pmc FETCH_SIZE   (separate passes, as MI355X_MICROARCH.md prescribes)
#   write/   --kernel-trace --pmc WRITE_SIZE
#   sq/      --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
#            (the calibrated VALU-busy measurement, profiles/r02/valu_calibration.md)
# Usage: scripts/profile_bench.sh <tag> [bench args...]     then: python scripts/summarize_profiles.py <tag> <version> [--masked]
set -u
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
export BTBA_BENCH_NPROC=1
ARGS="--steps 3 --warmup 1 --distinct 4 --no-cpu-baseline $*"
# the stats pass runs bench.py's DEFAULT step counts (20 timed + 3 warm-up), so that the kernel durations are taken at the
# same clocks as the bench line they are compared with; only the CPU baseline is left out (it spawns worker processes)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python "$REPO/bench.py" --no-cpu-baseline --distinct 8 $* > "$OUT/stats.log" 2>&1
echo "stats rc=$?" >> "$OUT/stats.log"
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o bench -- python "$REPO/bench.py" $ARGS --no-kernel-timing > "$OUT/fetch.log" 2>&1
echo "fetch rc=$?" >> "$OUT/fetch.log"
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o bench -- python "$REPO/bench.py" $ARGS --no-kernel-timing > "$OUT/write.log" 2>&1
echo "write rc=$?" >> "$OUT/write.log"
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -o bench -- python "$REPO/bench.py" $ARGS --no-kernel-timing > "$OUT/sq.log" 2>&1
echo "sq rc=$?" >> "$OUT/sq.log"
# keep the merge small: drop the raw per-dispatch kernel traces of the PMC passes (the counter CSVs carry the kernel names)
find "$OUT" -name "*kernel_trace.csv" -not -path "*stats*" -delete
find "$OUT" -type f | head -50
du -sh "$OUT"
