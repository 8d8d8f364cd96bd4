#!/bin/bash
# round 3, GPU call 35: v31a (unconditional prefetch) / v31b (+ clamp bounds as kernel arguments = scalar operands) / v31c (+ blend FMAs with the destination outside their sources)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_35
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 700 python scripts/ab_libs.py build/ab/v31a.so build/ab/v31b.so build/ab/v31c.so build/ab/v31a.so build/ab/v31b.so build/ab/v31c.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
