#!/bin/bash
# round 3, GPU call 57: scheduler settings, three alternations each
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_57
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 1100 python scripts/ab_libs.py build/ab/v31.so build/ab/fl_bias0.so build/ab/fl_relax.so build/ab/fl_ilpstrat.so build/ab/v31.so build/ab/fl_bias0.so build/ab/fl_relax.so build/ab/fl_ilpstrat.so build/ab/v31.so build/ab/fl_bias0.so build/ab/fl_relax.so build/ab/fl_ilpstrat.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
