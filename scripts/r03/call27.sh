#!/bin/bash
# round 3, GPU call 27: scripts/profile_bench.sh on the default workload (validation of the new per-workload profile pipeline)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$REPO"
bash scripts/profile_bench.sh r03_c3 2>&1 | tail -5
ls -la gpurun_out/prof_r03_c3/*/ | head -40; tail -3 gpurun_out/prof_r03_c3/*.log
