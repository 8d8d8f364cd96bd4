#!/bin/bash
# round 3, GPU call 25: defaults with the sparse items at the end; tile counts and c4 / c2 / masked / latency re-checked under it
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_25
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
export BTBA_BENCH_CACHE=/tmp/bench_inst.pkl
for v in "default:--latency" "tiles3:" "tiles4:" "tail128:" ; do
  name=${v%%:*}; args=${v#*:}
  case $name in tiles3) export BTBA_BENCH_TILES=3;; tiles4) export BTBA_BENCH_TILES=4;; tail128) unset BTBA_BENCH_TILES; export BTBA_SPARSE_TAIL=128;; *) unset BTBA_BENCH_TILES;; esac
  timeout 300 python bench.py --no-cpu-baseline $args > "$O/bench_$name.json" 2> "$O/bench.err"
  python - "$O/bench_$name.json" $name <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["frac"], round(1e3*j["roofline"]["avg_launch_ms"],2), j.get("single_instance"))
PY
done
unset BTBA_SPARSE_TAIL BTBA_BENCH_TILES BTBA_BENCH_CACHE
for v in "masked:--masked" "c2:--config c2" "c4:--config c4 --steps 40" "c4_tail0:--config c4 --steps 40"; do
  name=${v%%:*}; args=${v#*:}
  case $name in c4_tail0) export BTBA_SPARSE_TAIL=0;; *) unset BTBA_SPARSE_TAIL;; esac
  timeout 400 python bench.py --no-cpu-baseline $args > "$O/bench_$name.json" 2> "$O/bench.err"
  python - "$O/bench_$name.json" $name <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["bound"], j["roofline"]["frac"], round(1e3*j["roofline"]["avg_launch_ms"],2))
PY
done
