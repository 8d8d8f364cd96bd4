#!/bin/bash
# Round 6, GPU call 12: the 300-window fuzz and the session record on the FINAL build (hash 0ead505601e9ec3b).
OUT=gpurun_out/r06; mkdir -p $OUT; rm -f $OUT/session_record_final.jsonl
BTBA_SESSION_RECORD=$OUT/session_record_final.jsonl timeout 900 python -m pytest tests/test_tracking_session.py -q -m gpu -s -k "hip_vs_oracle" 2>&1 | grep -i "BA calls\|passed\|failed" | cut -c1-300
timeout 3300 python tests/tools/fuzz_parity.py 300 > $OUT/fuzz_parity_300_final.jsonl 2> $OUT/fuzz300.err; tail -1 $OUT/fuzz_parity_300_final.jsonl
