#!/bin/bash
OUT=gpurun_out/r06_final; mkdir -p $OUT/bench_lines
timeout 300 python bench.py --config c2 --no-cpu-baseline --parity --no-tracker-call > $OUT/bench_lines/bench_c2.json 2>/dev/null; cut -c1-200 $OUT/bench_lines/bench_c2.json; python -c "
import json; d=json.loads(open('gpurun_out/r06_final/bench_lines/bench_c2.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['traffic'], d['roofline']['valu_issue'], d['parity']['ok'])"
