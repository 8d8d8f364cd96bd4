#!/bin/bash
# round 3, GPU call 58: single-instance latency against tile / chunk counts on build v31
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_58
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 200 python scripts/latency_tiles.py > "$O/latency_tiles.jsonl" 2> "$O/err.txt"; cat "$O/latency_tiles.jsonl"; tail -3 "$O/err.txt"
