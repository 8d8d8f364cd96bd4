#!/bin/bash
# Round 6, GPU call 9: parity records beyond the campaign -- 300 fuzz windows against the reference and its self-spread, the c1 session's per-box record.
OUT=gpurun_out/r06; mkdir -p $OUT
BTBA_SESSION_RECORD=$OUT/session_record.jsonl timeout 900 python -m pytest tests/test_tracking_session.py -q -m gpu -s -k "hip_vs_oracle" 2>&1 | grep -i "BA call\|passed\|failed" | cut -c1-300
cat $OUT/session_record.jsonl | cut -c1-1500
timeout 3300 python tests/tools/fuzz_parity.py 300 > $OUT/fuzz_parity_300.jsonl 2> $OUT/fuzz300.err; tail -1 $OUT/fuzz_parity_300.jsonl
