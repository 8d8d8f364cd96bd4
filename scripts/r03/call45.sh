#!/bin/bash
# round 3, GPU call 45: timing probe (wrong data): 24-byte correspondences fetched as three 16-byte loads per lane and pair of entries instead of six 8-byte ones; c2 (sparse only)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_45
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
for rep in 1 2; do
for lib in v31 probe_c24f4; do
  BTBA_LIB_PATH=$REPO/build/ab/$lib.so timeout 300 python bench.py --config c2 --no-cpu-baseline 2> "$O/err_$lib.txt" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('$lib', j['value'], j['ms_per_step'], r.get('frac'), r.get('avg_launch_ms'))" | tee -a "$O/lines.txt"
done; done
