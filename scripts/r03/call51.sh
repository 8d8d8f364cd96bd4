#!/bin/bash
# round 3, GPU call 51: build() followed by smoke() in ONE process (libbtba.so must not become the process's HIP runtime before torch's)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r03_51
mkdir -p "$O"
export TMPDIR=/tmp
cd "$REPO"
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee "$O/smoke_main.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -2 | tee "$O/smoke_import.txt"
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('__BUILD_SMOKE_OK__')" 2>&1 | tail -2 | tee "$O/build_smoke.txt"
