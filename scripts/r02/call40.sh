#!/bin/bash
# round 2, GPU call 40: the other bench configurations on the final build (v21): masked frames, single-instance latency, c2, c4
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r02_40
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 200 python bench.py --masked --no-cpu-baseline > $O/bench_masked.json 2> $O/bench_masked.err
timeout 200 python bench.py --latency --no-cpu-baseline > $O/bench_latency.json 2> $O/bench_latency.err
timeout 200 python bench.py --config c2 --no-cpu-baseline --distinct 8 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --config c4 --no-cpu-baseline --distinct 4 --steps 10 > $O/bench_c4.json 2> $O/bench_c4.err
for f in masked latency c2 c4; do python -c "
import json,sys
d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d.get('single_instance'), d['roofline'].get('frac'))"; done
