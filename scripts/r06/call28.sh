#!/bin/bash
# Round 6, GPU call 28: the launch gaps of ONE masked window (the tracker's call): rocprofv3 kernel trace of 230 back-to-back solves, start / end of every kernel.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06/trace_b1; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o b1 -- python $GRAFT_REPO_ROOT/scripts/single_instance_trace.py --masked > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python - <<'PY'
import csv, glob, os, json, statistics as st
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06/trace_b1/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
rows = rows[len(rows) // 2:]            # steady state
dur, gap = {}, {}
for a, b in zip(rows, rows[1:]):
    dur.setdefault(a[2], []).append(a[1] - a[0]); gap.setdefault(a[2] + " -> " + b[2], []).append(b[0] - a[1])
out = {"kernel_ns_median": {k: st.median(v) for k, v in dur.items()}, "gap_ns_median (end of one kernel -> start of the next)": {k: [st.median(v), len(v)] for k, v in gap.items() if len(v) > 20}}
print(json.dumps(out, indent=1))
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06/launch_gaps_b1.json", "w"), indent=1)
PY
find $OUT -name "*.csv" -size +2M -delete
