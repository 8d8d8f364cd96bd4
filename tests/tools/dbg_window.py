"""Debug: one random window of tests/test_gpu_vs_reference.py::test_random_windows..., HIP (both cache layouts) vs oracle vs reference per iterate."""
import os, sys
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(_TESTS)); sys.path.insert(0, _TESTS)
import numpy as np, torch
from bundletrack_amd import synthetic as S
from bundletrack_amd.optimizer import BatchSolver, Workspace
from oracle import oracle as O, reference as R

def main(want):
    rng = np.random.default_rng(2024)
    for trial in range(20):
        K = int(rng.integers(2, 10)); m = int(rng.choice([0, 40, 150, 400])); bg = bool(rng.integers(0, 2)); wd = float(rng.choice([0.0, 1.0, 1.0]))
        if m == 0 and wd == 0.0: wd = 1.0
        pd_, pm = float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.001, 0.008))
        if trial != want: continue
        pb = S.make_problem(K, m, 5000 + trial, background=bg, full_res=False, perturb_deg=pd_, perturb_m=pm)
        campos, normals, intr = S.analytic_cache(pb)
        tr = O.solve(campos, normals, intr, pb.corr, pb.poses_init, params=O.default_params(weight_dense_depth=wd))
        dev = torch.device("cuda:0"); ws = Workspace()
        for layout in ("float4", "zn"):
            bs = BatchSolver(ws, weight_dense_depth=wd)
            corr, offs, mx = bs.pack_correspondences([pb.corr], K)
            corr_d = torch.from_numpy(corr.view(np.uint8).reshape(1, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
            poses_d = torch.from_numpy(pb.poses_init[None].copy()).to(dev)
            if layout == "float4":
                t = bs.solve(torch.from_numpy(campos[None]).to(dev), torch.from_numpy(normals[None]).to(dev), intr, corr_d, offs_d, mx, poses_d, trace=True)
            else:
                zn = S.compact_cache(pb)
                t = bs.solve_zn(torch.from_numpy(zn[None]).to(dev), pb.H, pb.W, pb.K, corr_d, offs_d, mx, poses_d, trace=True)
            tv = bs.trace_view(t)
            P = K * (K - 1) // 2
            for it in range(7):
                e = max(max(S.pose_error(tv.T_after[0, it, k], tr.T_after[it][k])) for k in range(K))
                cnt = tv.dense_pair[0, it, :, 27].astype(np.int64)
                print(layout, "it", it, f"pose diff vs oracle {e:.2e}", "count diff", np.abs(cnt - tr.dense_count[it][:P]).max(), "alpha hip", np.round(tv.pcg_scalars[0, it, :, 1], 4), "ora", np.round(tr.pcg_scalars[it][:, 1], 4))

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 9)
