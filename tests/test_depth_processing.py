"""SURVEY.md 8(f) rank 3 -- Frame::processDepth and Frame::depthToCloudAndNormals: CPU pins of the oracle
(not gpu) and HIP-vs-oracle parity (gpu)."""
import numpy as np
import pytest

from bundletrack_amd import synthetic as S


def noisy_depth(seed=0, H=96, W=128, holes=True):
    """Ellipsoid-on-background depth with sensor-like noise, flying pixels and holes."""
    rng = np.random.default_rng(seed)
    K = S.NOCS_K * np.array([[W / 640.0], [H / 480.0], [1.0]])
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    d, _ = S.render(S.orbit_pose(0.3), K, xs, ys, True)
    d = d + rng.normal(scale=0.0008, size=d.shape).astype(np.float32)
    if holes:
        d[rng.random(d.shape) < 0.03] = 0                       # missing returns
        fl = rng.random(d.shape) < 0.01
        d[fl] += rng.uniform(0.02, 0.2, size=int(fl.sum())).astype(np.float32)   # flying pixels
        d[:5] = 0; d[:, -7:] = 0
    return d.astype(np.float32), K.astype(np.float32)


def test_erode_semantics(oracle):
    d = np.full((7, 9), 0.5, np.float32)
    d[3, 4] = 0.6                                   # isolated outlier: all 8 neighbours differ by > diff -> 8/9 >= 0.8 -> removed
    d[0, 0] = 0.05                                  # below 0.1: zeroed outright
    out = oracle.erode_depth(d, 1, 0.001, 0.8)
    assert out[3, 4] == 0 and out[0, 0] == 0
    assert out[3, 3] == 0.5                         # its neighbours see only one bad pixel (1/9): kept
    assert out[6, 8] == 0.5                         # corner: out-of-image neighbours are NOT counted, denominator stays 9
    flat = oracle.erode_depth(np.full((5, 5), 0.4, np.float32))
    assert np.array_equal(flat, np.full((5, 5), 0.4, np.float32))


def test_gauss_filter_semantics(oracle):
    d = np.full((9, 9), 0.5, np.float32)
    out = oracle.gauss_filter_depth(d, 2, 2.0, 100000.0)
    assert np.allclose(out, 0.5, atol=1e-7)         # constant image is a fixed point, borders included
    d2 = d.copy(); d2[4, 4] = 0                      # a hole is FILLED from its neighbours (the centre need not be valid)
    assert abs(oracle.gauss_filter_depth(d2)[4, 4] - 0.5) < 1e-6
    d3 = d.copy(); d3[4, 4] = 0.53                   # |c - mean| >= 0.01: excluded from its own window, replaced by the rest
    assert abs(oracle.gauss_filter_depth(d3)[4, 4] - 0.5) < 1e-6
    assert np.all(oracle.gauss_filter_depth(np.zeros((6, 6), np.float32)) == 0)


def test_normals_point_to_camera_and_match_analytic(oracle):
    H, W = 120, 160
    K = (S.NOCS_K * np.array([[0.25], [0.25], [1.0]])).astype(np.float32)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    d, n_true = S.render(S.orbit_pose(0.2), K.astype(np.float64), xs, ys, False)
    n, xyz = oracle.depth_to_normals(d, K)
    valid = np.linalg.norm(n[..., :3], axis=-1) > 0
    assert valid.sum() > 100 and np.all(n[..., 3] == 0)
    assert np.all(n[~(d >= 0.1)] == 0) and np.all(n[0] == 0) and np.all(n[:, 0] == 0)       # invalid depth and image border
    assert np.allclose(np.linalg.norm(n[valid][:, :3], axis=1), 1, atol=1e-5)
    assert np.all((n[valid][:, :3] * -xyz[valid][:, :3]).sum(1) >= 0)                        # oriented towards the camera
    inner = valid & (np.linalg.norm(n_true[..., :3], axis=-1) > 0)
    cosang = (n[inner][:, :3] * n_true[inner][:, :3]).sum(1)
    assert np.median(cosang) > 0.99                  # finite-difference normals agree with the analytic surface normal
    assert np.array_equal(xyz[..., 2][d >= 0.1], d[d >= 0.1])


def test_process_depth_is_erode_then_two_filter_passes(oracle):
    d, _ = noisy_depth(1)
    a = oracle.process_depth(d)
    b = oracle.gauss_filter_depth(oracle.gauss_filter_depth(oracle.erode_depth(d)))
    assert np.array_equal(a, b)
    assert (a > 0).sum() > 0.5 * d.size


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(480, 640), (96, 128), (37, 53)])
def test_gpu_process_depth_matches_oracle(oracle, shape):
    import torch
    from bundletrack_amd.optimizer import Workspace, process_depth
    ws = Workspace()
    d, _ = noisy_depth(2, *shape)
    got = process_depth(ws, torch.from_numpy(d).cuda()).cpu().numpy()
    ref = oracle.process_depth(d)
    # the erode stage and every validity / mean-gate decision are comparisons -> identical zero pattern, except where
    # |c - mean| sits within an ulp of the 0.01 gate; values differ only through expf ulps and FMA contraction
    mism = (got == 0) != (ref == 0)
    assert mism.sum() <= 2
    ok = ~mism
    assert np.abs(got[ok] - ref[ok]).max() < 2e-6
    # other parameters (bigger stencils, tight range sigma)
    got2 = process_depth(ws, torch.from_numpy(d).cuda(), erode_radius=2, erode_ratio=0.5, bf_radius=3, sigma_d=1.5, sigma_r=0.01).cpu().numpy()
    ref2 = oracle.process_depth(d, 2, 0.001, 0.5, 3, 1.5, 0.01)
    mism2 = (got2 == 0) != (ref2 == 0)
    assert mism2.sum() <= 2 and np.abs(got2[~mism2] - ref2[~mism2]).max() < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(480, 640), (61, 67)])
def test_gpu_normals_match_oracle(oracle, shape):
    import torch
    from bundletrack_amd.optimizer import Workspace, depth_to_normals
    ws = Workspace()
    d, K = noisy_depth(3, *shape)
    d = oracle.process_depth(d)
    n, xyz = depth_to_normals(ws, torch.from_numpy(d).cuda(), K, want_xyz=True)
    n, xyz = n.cpu().numpy(), xyz.cpu().numpy()
    rn, rxyz = oracle.depth_to_normals(d, K)
    assert np.array_equal(xyz.view(np.uint32), rxyz.view(np.uint32))             # back-projection: bit-exact
    assert np.array_equal(np.linalg.norm(n, axis=-1) == 0, np.linalg.norm(rn, axis=-1) == 0)     # same validity decisions
    assert np.abs(n - rn).max() < 2e-6                                            # IEEE sqrt/div, contraction off


@pytest.mark.gpu
def test_gpu_frame_pipeline_feeds_the_optimiser(oracle):
    """processDepth -> normals -> optimizeFrames end to end on the device; same poses as the all-oracle chain."""
    import torch
    from bundletrack_amd.optimizer import OptimizerGpu, Workspace, depth_to_normals, process_depth
    ws = Workspace()
    pb = S.make_problem(3, 200, seed=77, background=False)
    rng = np.random.default_rng(5)
    raw = pb.depth + (pb.depth > 0) * rng.normal(scale=0.0005, size=pb.depth.shape).astype(np.float32)
    dep_g, nrm_g, dep_o, nrm_o = [], [], [], []
    for k in range(3):
        dg = process_depth(ws, torch.from_numpy(raw[k]).cuda())
        dep_g.append(dg); nrm_g.append(depth_to_normals(ws, dg, pb.K))
        do = oracle.process_depth(raw[k]); dep_o.append(do); nrm_o.append(oracle.depth_to_normals(do, pb.K)[0])
    poses = pb.poses_init.copy()
    OptimizerGpu(workspace=ws).optimizeFrames(pb.corr, pb.n_match_per_pair, 3, pb.H, pb.W, dep_g, None, nrm_g, poses, pb.K)
    caches = [oracle.build_cache(dep_o[k], nrm_o[k], pb.K) for k in range(3)]
    ref = oracle.solve(np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches]), caches[0]["intr"], pb.corr, pb.poses_init)
    for k in range(3):
        r, t = S.pose_error(poses[k], ref.poses[k])
        assert r < 1e-4 and t < 1e-4
