#!/bin/bash
# round 3, GPU call 61: 120 randomised windows (tests/tools/fuzz_parity.py), looking for an unexplained iterate
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_61
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 500 python tests/tools/fuzz_parity.py 120 > "$O/fuzz.jsonl" 2> "$O/err.txt"
python - "$O/fuzz.jsonl" <<'PY'
import json,sys
rows=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"case"')]
bad=[r for r in rows if r.get("unexplained_iterates") or not r["finite"]]
import statistics
print("cases", len(rows), "above 1e-4:", sum(max(r["rot"],r["trans"])>=1e-4 for r in rows), "median", statistics.median(max(r["rot"],r["trans"]) for r in rows), "unexplained / non-finite:", len(bad))
for r in bad: print(json.dumps(r)[:600])
PY
tail -2 "$O/err.txt"
