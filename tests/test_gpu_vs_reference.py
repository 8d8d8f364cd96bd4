"""The HIP path against THE REFERENCE'S OWN SOLVER: wenbowen123/BundleTrack's solveBundlingStub and all its kernels,
compiled for the CPU in the build container (oracle/_ref/libbtba_ref_solver.so, see tests/test_oracle_vs_reference.py) and
shipped to the GPU box as a prebuilt file -- nothing under /root/reference is read here.  Skipped if the file is absent."""
import os

import numpy as np
import pytest

from bundletrack_amd import synthetic as S
from oracle import reference as R

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(R.SO_SOLVER), reason="oracle/_ref/libbtba_ref_solver.so not built")]


def hip_solve(pb, wd):
    import torch
    from bundletrack_amd.optimizer import BatchSolver, Workspace
    dev = torch.device("cuda:0")
    bs = BatchSolver(Workspace(), weight_dense_depth=wd)
    corr, offs, mx = bs.pack_correspondences([pb.corr], pb.n_frames)
    zn = np.concatenate([pb.cache_depth[..., None], pb.cache_normals[..., :3]], -1).astype(np.float32)
    zn_d = torch.from_numpy(zn[None]).to(dev)
    corr_d = torch.from_numpy(corr.view(np.uint8).reshape(1, -1, 32)).to(dev)
    offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
    poses_d = torch.from_numpy(pb.poses_init[None].copy()).to(dev)
    bs.solve_zn(zn_d, pb.H, pb.W, pb.K, corr_d, offs_d, mx, poses_d)
    return poses_d.cpu().numpy()[0]


@pytest.mark.parametrize("name,K,m,wd,seed,bg", [
    ("c2", 10, 1000, 0.0, S.config_seed(2), True),       # BASELINE configs[1]
    ("c3", 15, 2000, 1.0, S.config_seed(3), True),       # BASELINE configs[2]: the headline configuration
    ("c3-masked", 15, 2000, 1.0, S.config_seed(3), False),
    ("window", 5, 300, 1.0, 28, False),
])
def test_hip_matches_the_reference_solver(name, K, m, wd, seed, bg):
    pb = S.make_problem(K, m, seed, background=bg, full_res=False)
    campos, normals, intr = S.analytic_cache(pb)
    ref, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, weight_dense=wd)
    got = hip_solve(pb, wd)
    worst = max(max(S.pose_error(got[k], ref[k])) for k in range(K))
    print(f"{name}: HIP vs the reference's own solver after 7 GN x 5 PCG: worst pose difference {worst:.2e}")
    assert worst < 1e-4, worst
