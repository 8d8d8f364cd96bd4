#!/bin/bash
# round 2, GPU call 23: block ranges as part of the frame cache (btba_zn_block_ranges), k_system_solve phase stamps at B = 32
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$REPO/gpurun_out/r02_23
mkdir -p "$O"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_tracking_session.py tests/test_cpp_host.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
timeout 200 python scripts/sys_clocks.py > $O/sys_clocks.txt 2>&1; tail -2 $O/sys_clocks.txt
