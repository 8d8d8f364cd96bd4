#!/bin/bash
# round 3, GPU call 63: SURVEY 8(f) rows re-measured on build v32: per-frame depth pre-processing + normals, frame cache build (inside boundary_timing), RANSAC
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_63
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 200 python tests/tools/image_timing.py > "$O/image_timing_v32.json" 2> "$O/err.txt"; cat "$O/image_timing_v32.json" | cut -c1-1500; tail -2 "$O/err.txt"
