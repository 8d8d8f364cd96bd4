#!/bin/bash
# round 3, GPU call 11: wave priority experiments (prologue / epilogue high, pixel loop low; sparse workgroups high)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_11
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 800 python scripts/ab_libs.py build/ab/r03g.so build/ab/r03i.so build/ab/r03j.so build/ab/r03k.so build/ab/r03l.so build/ab/r03g.so build/ab/r03i.so build/ab/r03j.so build/ab/r03k.so build/ab/r03l.so > "$O/ab.jsonl" 2> "$O/ab.err"; cat "$O/ab.jsonl"; tail -3 "$O/ab.err"
