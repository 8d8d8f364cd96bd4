#!/usr/bin/env python
"""Per-kernel means of the counters collected by scripts/pmc_memory_path.sh.

    python scripts/summarize_pmc.py gpurun_out/pmc_<tag> <out.json>

Reads every *counter_collection.csv under the directory (one per rocprofv3 --pmc pass), averages each counter over the dispatches of each
kernel (the first dispatch of a kernel is dropped: cold caches) and writes {kernel: {counter: mean per launch, "launches": n}}."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root, out = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        per_dispatch = collections.defaultdict(dict)
        with open(path) as f:
            for row in csv.DictReader(f):
                kern = row.get("Kernel_Name", "")
                per_dispatch[(kern, row.get("Dispatch_Id"))][row["Counter_Name"]] = per_dispatch[(kern, row.get("Dispatch_Id"))].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        seen = set()
        for (kern, did), vals in sorted(per_dispatch.items(), key=lambda kv: int(kv[0][1])):
            if kern not in seen:
                seen.add(kern)          # cold first launch
                continue
            for c, v in vals.items():
                acc[kern][c].append(v)
    res = {}
    for kern, counters in acc.items():
        short = kern.split("(")[0]
        if not any(k in short for k in ("k_fused_sweeps", "k_system_solve", "k_solve_small", "k_sparse_sweep", "k_dense_sweep", "k_big_")):
            continue
        res[short] = {c: round(sum(v) / len(v), 1) for c, v in sorted(counters.items())}
        res[short]["launches"] = max(len(v) for v in counters.values())
    args = open(os.path.join(root, "args.txt")).read().strip() if os.path.exists(os.path.join(root, "args.txt")) else ""
    json.dump({"what": "mean per launch, summed over the chip (rocprofv3 --kernel-trace --pmc, one pass per counter set: scripts/pmc_memory_path.sh); " + args, "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
