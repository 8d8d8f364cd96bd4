#!/bin/bash
# round 3, GPU call 50: what the driver runs at round end, on the committed tree: build check, GPU suite, smoke(), default bench
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_50
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee "$O/pytest_gpu.txt"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee "$O/smoke.txt"
timeout 600 python bench.py > "$O/bench.json" 2> "$O/bench.err"; python - "$O/bench.json" <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]; c=j["cpu_baseline"]
print(j["value"], j["ms_per_step"], r["bound"], r["frac"], r["avg_launch_ms"], r["traffic"], (r.get("valu_issue") or {}).get("busy_frac"), c["value"], c["cores"], c["kind"])
PY
