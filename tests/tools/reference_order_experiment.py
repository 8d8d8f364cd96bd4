#!/usr/bin/env python
"""Is the ORDER of the per-pixel arithmetic what makes the HIP path's accept decisions differ from the oracle's?  (round 4's verdict, "prove the
arithmetic-order explanation".)  GPU box, one library per process:

    hipcc ... -DBTBA_REFERENCE_ORDER -ffp-contract=off -o build/ab/reforder.so bundletrack_amd/csrc/btba_api.hip        (a TEST-ONLY build, never the product)
    BTBA_LIB_PATH=build/ab/reforder.so python tests/tools/reference_order_experiment.py 120 > gpurun_out/reforder.jsonl
                                       python tests/tools/reference_order_experiment.py 120 > gpurun_out/product.jsonl

The experiment build evaluates every pixel of the float4-cache dense sweep (the path the traced fuzz windows run) as the reference does --
findDenseCorr / bilinearInterpolationFloat4 statement by statement, IEEE division and square root, no fused multiply-add contraction anywhere in the
library (btba_kernels.hpp: pixel_reference_order).  Per window of tests/tools/fuzz_parity.py: the (iterate, dense pair) cells whose accepted-pixel
count differs from the oracle's, and -- the clean case -- the cells of the FIRST iterate, where both sides start from the same poses.  "The same poses"
is checked, not assumed: the window's starting T and T^-1 as the device computes them (Log -> Exp -> generic inverse, device libm) against the oracle's
(host libm), bit for bit; a first-iterate cell that still differs is attributed to a frame whose matrices differ in the last bits, or reported as
unexplained."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

import numpy as np  # noqa: E402


def main():
    import torch
    import fuzz_parity as F
    from bundletrack_amd import _lib
    from bundletrack_amd.optimizer import Workspace
    from oracle import oracle as O
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    ws = Workspace()
    dev = torch.device("cuda:0")

    def device_matrices(poses):
        N = poses.shape[0]
        P = torch.from_numpy(np.ascontiguousarray(poses, np.float32).reshape(N, 16)).to(dev)
        x = torch.zeros((N, 6), device=dev); T = torch.zeros((N, 16), device=dev); Ti = torch.zeros((N, 16), device=dev)
        L = _lib.lib()
        _lib.check(L.btba_matrices_to_poses(ws.handle, N, P.data_ptr(), x.data_ptr()), "m2p")
        _lib.check(L.btba_poses_to_matrices(ws.handle, N, x.data_ptr(), T.data_ptr(), Ti.data_ptr()), "p2m")
        ws.sync()
        return T.cpu().numpy().reshape(N, 4, 4), Ti.cpu().numpy().reshape(N, 4, 4)

    def hook(rec, pb, corr, caches, ref, tv):
        hip = np.rint(tv.dense_pair[0][..., 27]).astype(np.int64)             # [iterate, pair] accepted pixels
        ora = np.asarray(ref.dense_count, np.int64)[:, :hip.shape[1]]
        d = hip[:ora.shape[0]] - ora
        rec["count_cells"] = int(d.size)
        rec["count_cells_differing"] = int((d != 0).sum())
        rec["count_pixels_differing"] = int(np.abs(d).sum())
        first = np.nonzero(d[0] != 0)[0] if d.shape[0] else np.zeros(0, np.int64)
        rec["first_iterate_cells"] = int(d.shape[1]) if d.shape[0] else 0
        rec["first_iterate_cells_differing"] = int(first.size)
        # the starting matrices on both sides
        K = pb.n_frames
        Th, Tih = device_matrices(pb.poses_init)
        To = np.stack([O.pose_to_matrix(*O.matrix_to_pose(pb.poses_init[k])) for k in range(K)]).astype(np.float32)
        Tio = np.stack([O.mat4_inverse(To[k]) for k in range(K)]).astype(np.float32)
        if keep_T:                              # both sides take the given matrix as T; the inverses are formed from it on each side (same operation order)
            Th = To = np.asarray(pb.poses_init, np.float32).copy()
            Tih = Tio = np.stack([O.mat4_inverse(Th[k]) for k in range(K)]).astype(np.float32)
        bad = [k for k in range(K) if not (np.array_equal(Th[k].view(np.uint32), To[k].view(np.uint32)) and np.array_equal(Tih[k].view(np.uint32), Tio[k].view(np.uint32)))]
        rec["frames_with_other_start_matrices"] = bad
        if first.size:
            pairs = [(i, j) for i in range(K) for j in range(i + 1, K)]
            rec["first_iterate_differing_pairs"] = [[int(pairs[p][0]), int(pairs[p][1]), int(d[0, p])] for p in first]
            rec["first_iterate_cells_differing_with_identical_matrices"] = int(sum(1 for p in first if pairs[p][0] not in bad and pairs[p][1] not in bad))

    keep_T = bool(os.environ.get("BTBA_PREPARE_KEEP_T"))
    if keep_T:
        os.environ["ORC_KEEP_T"] = "1"          # the oracle's first-iterate T is the given matrix as well (oracle/btba_oracle.c; experiment only)

    tot = {"cases": 0, "above_1e-4": 0, "cells": 0, "cells_differing": 0, "pixels_differing": 0, "first_iterate_cells": 0, "first_iterate_cells_differing": 0,
           "first_iterate_cells_differing_with_identical_matrices": 0, "windows_with_other_start_matrices": 0, "unexplained": []}
    for rec in F.run_cases(n, explain_always=True, hook=hook):
        print(json.dumps(rec), flush=True)
        tot["cases"] += 1
        tot["above_1e-4"] += int(max(rec["rot"], rec["trans"]) >= 1e-4)
        for k in ("cells", "cells_differing", "pixels_differing"):
            tot[k] += rec.get("count_" + k, 0)
        tot["first_iterate_cells"] += rec.get("first_iterate_cells", 0)
        tot["first_iterate_cells_differing"] += rec.get("first_iterate_cells_differing", 0)
        tot["first_iterate_cells_differing_with_identical_matrices"] += rec.get("first_iterate_cells_differing_with_identical_matrices", 0)
        tot["windows_with_other_start_matrices"] += int(bool(rec.get("frames_with_other_start_matrices")))
        if rec.get("unexplained_iterates"):
            tot["unexplained"].append(rec["case"])
    tot["library"] = os.path.basename(os.environ.get("BTBA_LIB_PATH", "libbtba.so (the product)"))
    tot["start_matrices"] = "the window's input matrices as they stand on BOTH sides (BTBA_PREPARE_KEEP_T / ORC_KEEP_T): bit-identical first-iterate T" if keep_T else "Exp(Log(P)) on each side (device libm / host libm)"
    print(json.dumps({"summary": tot}), flush=True)


if __name__ == "__main__":
    main()
