#!/bin/bash
# Profiles ONE bench workload on the GPU box with rocprofv3 -- every pass on the workload the bench line reports (same --config,
# --instances, --distinct (default 32), mask and layouts; only step counts differ).  Writes under gpurun_out/prof_<tag>/ :
#   stats/   --kernel-trace --stats            (per-kernel durations at bench.py's default step counts)
#   fetch/   --kernel-trace --pmc FETCH_SIZE   (separate passes, as MI355X_MICROARCH.md prescribes)
#   write/   --kernel-trace --pmc WRITE_SIZE
#   sq/      --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
#            (the calibrated VALU-busy measurement, profiles/r02/valu_calibration.md)
# Usage: scripts/profile_bench.sh <tag> [bench args, e.g. --masked | --config c4]     then: python scripts/summarize_profiles.py <tag> <version> [same bench args]
set -u
TAG=${1:-r04}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
export BTBA_BENCH_CACHE=/tmp/bench_instances_$TAG.npz      # the passes share one set of generated instances
echo "bench args: $*" > "$OUT/args.txt"
# instances generated once, by a plain run (child processes allowed), before any profiler is attached
BTBA_BENCH_NPROC=8 timeout 300 python "$REPO/bench.py" --no-cpu-baseline --steps 2 --warmup 1 --settle-ms 0 $* > "$OUT/pre.log" 2>&1
echo "pre rc=$?" >> "$OUT/pre.log"
export BTBA_BENCH_NPROC=1
# the stats pass runs bench.py's DEFAULT step counts, so that the kernel durations are taken at the same clocks as the bench line
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python "$REPO/bench.py" --no-cpu-baseline --no-tracker-call --no-incl-pack --no-single-instance $* > "$OUT/stats.log" 2>&1
echo "stats rc=$?" >> "$OUT/stats.log"
ARGS="--steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-tracker-call --no-incl-pack --no-single-instance --no-kernel-timing $*"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o bench -- python "$REPO/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
echo "fetch rc=$?" >> "$OUT/fetch.log"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o bench -- python "$REPO/bench.py" $ARGS > "$OUT/write.log" 2>&1
echo "write rc=$?" >> "$OUT/write.log"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -o bench -- python "$REPO/bench.py" $ARGS > "$OUT/sq.log" 2>&1
echo "sq rc=$?" >> "$OUT/sq.log"
# keep the merge small: drop the raw per-dispatch kernel traces (the counter CSVs carry the kernel names; the stats CSV is the summary)
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
rm -f "$BTBA_BENCH_CACHE"
du -sh "$OUT"
