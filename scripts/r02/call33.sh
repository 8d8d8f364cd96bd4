#!/bin/bash
# round 2, GPU call 33: large windows -- reduction and assembly on many workgroups
mkdir -p gpurun_out/r02_33
O=gpurun_out/r02_33
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference.py -m gpu -x -q -s -k "edge_shapes or N_40 or n40 or large or reference" > $O/pytest.log 2>&1; grep -n "N=\|passed\|failed\|Error" $O/pytest.log | cut -c1-200 | tail -14
timeout 300 python scripts/large_window_timing.py > $O/large_window_timing.jsonl 2> $O/lw.err; cut -c1-420 $O/large_window_timing.jsonl | tail -4
BTBA_NO_BIG_ASSEMBLY=1 timeout 300 python scripts/large_window_timing.py > $O/large_window_timing_one_wg.jsonl 2>> $O/lw.err; cut -c1-260 $O/large_window_timing_one_wg.jsonl | tail -3
