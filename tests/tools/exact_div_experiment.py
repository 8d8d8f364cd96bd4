#!/usr/bin/env python
"""How much of the HIP path's disagreement with the oracle's DECISIONS is due to its 1-ulp reciprocal / reciprocal square root (v_rcp_f32, v_rsq_f32:
the counterpart of the reference's own -use_fast_math, CMakeLists.txt:7) and how much to its re-ordered arithmetic?  GPU box, one library per process:

    BTBA_LIB_PATH=build/ab/exactdiv.so python tests/tools/exact_div_experiment.py 120 > gpurun_out/exact_div_exact.jsonl
                                       python tests/tools/exact_div_experiment.py 120 > gpurun_out/exact_div_product.jsonl

Runs the 120 windows of tests/tools/fuzz_parity.py with both traces for EVERY case and prints per case: the final pose difference, the first
decision the two sides take differently (tests/helpers.py: an accepted-pixel count of a dense pair, or a PCG epsilon guard), the number of
(iterate, dense pair) cells whose accepted-pixel counts differ and by how many pixels, and the per-iterate differences next to the oracle's own
summation-order spread."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

import numpy as np  # noqa: E402


def main():
    import fuzz_parity as F
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    cells = {}

    def hook(rec, pb, corr, caches, ref, tv):
        hip = np.rint(tv.dense_pair[0][..., 27]).astype(np.int64)             # [iterate, pair] accepted pixels
        ora = np.asarray(ref.dense_count, np.int64)[:, :hip.shape[1]]
        d = hip[:ora.shape[0]] - ora
        rec["count_cells"] = int(d.size)
        rec["count_cells_differing"] = int((d != 0).sum())
        rec["count_pixels_differing"] = int(np.abs(d).sum())
        rec["first_iterate_cells_differing"] = int((d[0] != 0).sum()) if d.shape[0] else 0      # iterate 0: IDENTICAL inputs on both sides

    tot = {"cases": 0, "above_1e-4": 0, "cells": 0, "cells_differing": 0, "pixels_differing": 0, "first_iterate_cells_differing": 0, "unexplained": []}
    for rec in F.run_cases(n, explain_always=True, hook=hook):
        print(json.dumps(rec), flush=True)
        tot["cases"] += 1
        tot["above_1e-4"] += int(max(rec["rot"], rec["trans"]) >= 1e-4)
        tot["cells"] += rec.get("count_cells", 0); tot["cells_differing"] += rec.get("count_cells_differing", 0)
        tot["pixels_differing"] += rec.get("count_pixels_differing", 0); tot["first_iterate_cells_differing"] += rec.get("first_iterate_cells_differing", 0)
        if rec.get("unexplained_iterates"):
            tot["unexplained"].append(rec["case"])
    tot["library"] = os.path.basename(os.environ.get("BTBA_LIB_PATH", "libbtba.so (the product)"))
    print(json.dumps({"summary": tot}), flush=True)


if __name__ == "__main__":
    main()
