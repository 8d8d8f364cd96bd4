#!/bin/bash
# Round 6, GPU call 13: pinned I/O block for the boundary's small copies (poses in / out, pair offsets): tracker_call A/B.
timeout 900 python scripts/r06/tracker_ab.py build/ab/r06_fast.so build/ab/r06_pinio.so build/ab/r06_fast.so build/ab/r06_pinio.so build/ab/r06_fast.so build/ab/r06_pinio.so 2>&1 | tee gpurun_out/r06/tracker_ab.jsonl
