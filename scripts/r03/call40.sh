#!/bin/bash
# round 3, GPU call 40: block-list entry one trip ahead in a scalar register (no LDS round trip in front of the stream prefetch)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_40
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v31c.so build/ab/v31e.so build/ab/v31c.so build/ab/v31e.so build/ab/v31c.so build/ab/v31e.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
