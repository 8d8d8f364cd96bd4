"""Solver-level pins of the CPU oracle (all self-derived -- the reference has no fixtures):
exact-data fixed point, K=2 closed form, fp32 C oracle vs the independent fp64 numpy restatement,
frame cache formulas, regression against tests/golden/*.npz, thread-count independence."""
import glob
import os

import numpy as np
import pytest

from bundletrack_amd import synthetic as S
from oracle import oracle_np as ONP

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")) if not os.path.basename(p).startswith("ref_"))


def _cache(oracle, pb):
    caches = [oracle.build_cache(pb.depth[k], pb.normals[k], pb.K, pb.downscale) for k in range(pb.n_frames)]
    return np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches]), caches[0]["intr"], caches


def test_cache_formulas(oracle, small_problem):
    pb = small_problem
    campos, normals, intr, caches = _cache(oracle, pb)
    Hd, Wd = campos.shape[1:3]
    assert (Hd, Wd) == (120, 160)
    K = pb.K.astype(np.float64)
    # CUDACache.cpp:20-24
    want = [K[0, 0] * Wd / pb.W, K[1, 1] * Hd / pb.H, K[0, 2] * (Wd - 1) / (pb.W - 1), K[1, 2] * (Hd - 1) / (pb.H - 1)]
    assert np.allclose(intr, want, rtol=1e-6)
    xi, yi = S.cache_source_pixels(pb.H, pb.W, Hd, Wd)
    assert xi[0] == 0 and xi[-1] == pb.W - 1 and yi[-1] == pb.H - 1          # nearest-neighbour pick spans the image
    d = pb.depth[1][yi][:, xi]
    assert np.array_equal(caches[1]["depth"], d)
    assert np.array_equal(caches[1]["normals"], pb.normals[1][yi][:, xi])
    x = (xi[None, :] - K[0, 2]) / K[0, 0] * d
    y = (yi[:, None] - K[1, 2]) / K[1, 1] * d
    assert np.abs(caches[1]["campos"][..., 0] - x).max() < 2e-6 and np.abs(caches[1]["campos"][..., 1] - y).max() < 2e-6
    assert np.array_equal(caches[1]["campos"][..., 2], d) and np.all(caches[1]["campos"][..., 3] == 1)
    assert caches[1]["n_valid"] == int((d >= 0.1).sum())


def test_cache_invalid_depth(oracle):
    depth = np.zeros((8, 12), np.float32)
    depth[2:6, 3:9] = 0.5
    depth[3, 4] = 0.09          # below the 0.1 validity threshold (CUDAImageUtil.cu:320)
    normals = np.zeros((8, 12, 4), np.float32)
    normals[..., 2] = -1
    c = oracle.build_cache(depth, normals, S.NOCS_K.astype(np.float32), 2.0)
    assert c["campos"].shape == (4, 6, 4)
    inval = c["depth"] < 0.1
    assert np.all(c["campos"][inval] == 0) and np.all(c["campos"][~inval][:, 3] == 1)
    assert c["n_valid"] == int((~inval).sum())


def test_fixed_point_on_exact_data(oracle):
    """GT poses + noise-free correspondences => the sparse update is (numerically) zero.  The dense term is
    only approximately stationary there: bilinear lookups of a curved surface and across the silhouette are
    not exact geometry, so it keeps a small bias."""
    pb = S.make_problem(4, 120, seed=9, background=True, perturb_deg=0.0, perturb_m=0.0, noise_m=0.0, outlier_frac=0.0)
    campos, normals, intr, _ = _cache(oracle, pb)
    tr = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(weight_dense_depth=0.0))
    for it in range(7):
        assert np.abs(tr.delta[it]).max() < 2e-5
    for k in range(pb.n_frames):
        r, t = S.pose_error(tr.poses[k], pb.poses_gt[k])
        assert r < 5e-5 and t < 5e-5
    tr = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init)
    assert np.abs(tr.delta).max() < 1e-3
    for k in range(pb.n_frames):
        r, t = S.pose_error(tr.poses[k], pb.poses_gt[k])
        assert r < 2e-3 and t < 2e-3


def test_two_frame_sparse_closed_form(oracle):
    """K=2, sparse only, PCG run to convergence: one GN step equals the 6x6 linearised 3D-3D alignment."""
    pb = S.make_problem(2, 400, seed=10, background=True, outlier_frac=0.0, noise_m=0.0005)
    campos, normals, intr, _ = _cache(oracle, pb)
    prm = oracle.default_params(weight_dense_depth=0.0, n_gn_iters=1, n_pcg_iters=60, robust_delta=10.0)
    tr = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=prm)
    T0, T1 = pb.poses_init[0].astype(np.float64), pb.poses_init[1].astype(np.float64)
    wi = pb.corr["pos_i"].astype(np.float64) @ T0[:3, :3].T + T0[:3, 3]
    wj = pb.corr["pos_j"].astype(np.float64) @ T1[:3, :3].T + T1[:3, 3]
    r = wi - wj
    J = np.zeros((len(r), 3, 6))                       # d r / d(rot_1, trans_1) = -[D(w_j) | I] = [ [w_j]x | -I ]
    J[:, 0, 1], J[:, 0, 2] = -wj[:, 2], wj[:, 1]
    J[:, 1, 0], J[:, 1, 2] = wj[:, 2], -wj[:, 0]
    J[:, 2, 0], J[:, 2, 1] = -wj[:, 1], wj[:, 0]
    J[:, :, 3:] = -np.eye(3)
    J = J.reshape(-1, 6)
    delta = np.linalg.solve(J.T @ J, -J.T @ r.reshape(-1))
    # The absolute eps-guards (alpha = 0 once p^T A p <= 1e-6, SolverBundling.cu:757) freeze the iteration
    # before full convergence -- the "convergence floor" of SURVEY.md section 7 -- so the weakly observed
    # rotation only gets within ~1e-3 of the exact linear solve; translation converges.
    assert np.abs(tr.delta[0, 1, :3] - delta[:3]).max() < 1.5e-3
    assert np.abs(tr.delta[0, 1, 3:] - delta[3:]).max() < 2e-5
    assert np.all(tr.pcg_scalars[0, -10:, 1] == 0.0)          # the guard really tripped


@pytest.mark.parametrize("wd,bg", [(0.0, True), (1.0, True), (1.0, False)])
def test_c_oracle_matches_fp64_restatement(oracle, wd, bg):
    """Literal fp32 C (3x12*12x6 Jacobians, matrix-free J^T J) vs closed-form fp64 numpy (explicit normal
    matrix): different derivations of the same algorithm must agree per Gauss-Newton iterate."""
    pb = S.make_problem(4, 200, seed=12, background=bg)
    campos, normals, intr, _ = _cache(oracle, pb)
    tr = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(weight_dense_depth=wd))
    nr = ONP.solve(campos, normals, intr, pb.corr, pb.poses_init, weight_dense_depth=wd)
    tol = 2e-6 if wd == 0.0 else 1e-4      # dense: accept/reject flips of single pixels differ between fp32 and fp64
    for it in range(7):
        for k in range(pb.n_frames):
            r, t = S.pose_error(tr.T_after[it, k], nr["T_after"][it][k])
            assert r < tol and t < tol, (it, k, r, t)
        if wd > 0:
            assert np.abs(tr.dense_count[it][:6] - np.array(nr["dense_count"][it])).max() <= 3


def test_sparse_operator_matches_explicit_matrix(oracle):
    """PCGStep_Kernel0 + Kernel1a (matrix-free, unweighted J^T J) == the explicit block matrix built from
    per-pair moment sums (what the HIP path assembles)."""
    pb = S.make_problem(4, 150, seed=13, background=True)
    rng = np.random.default_rng(0)
    p = rng.normal(size=(4, 6)).astype(np.float32)
    p[0] = 0
    T = pb.poses_init
    got = oracle.sparse_apply(pb.corr, T, p)
    campos = np.zeros((4, 4, 4, 4), np.float32); normals = np.zeros((4, 4, 4, 4), np.float32)
    nr = ONP.solve(campos, normals, [1, 1, 0, 0], pb.corr, T, weight_dense_depth=0.0, n_gn_iters=1, n_pcg_iters=0)
    A = nr["A"][0]                                    # [trans, rot] per frame
    pv = np.concatenate([p[:, 3:], p[:, :3]], 1).reshape(-1)
    Ap = (A @ pv).reshape(4, 6)
    want = np.concatenate([Ap[:, 3:], Ap[:, :3]], 1)
    assert np.abs(got[1:] - want[1:]).max() < 2e-4 * np.abs(want).max()


def test_thread_count_and_accum_mode(oracle, small_problem):
    pb = small_problem
    campos, normals, intr, _ = _cache(oracle, pb)
    a = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(n_threads=1))
    b = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(n_threads=4))
    assert np.array_equal(a.poses, b.poses)              # double-carried sums: order-free at fp32 output precision
    c = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(n_threads=1, accum_mode=0))
    # Sequential fp32 summation of 19 200-term sums (one of the orders the reference's float atomics can
    # take) moves the result by up to ~1e-3 on this weakly conditioned K=4 problem: the reference itself is
    # only reproducible to that level, which is why the canonical oracle carries the sums in double.
    for k in range(pb.n_frames):
        r, t = S.pose_error(a.poses[k], c.poses[k])
        assert r < 5e-3 and t < 5e-3


def test_invalid_entries_and_empty_pairs(oracle, small_problem):
    pb = small_problem
    campos, normals, intr, _ = _cache(oracle, pb)
    corr = pb.corr.copy()
    corr["imgIdx_i"][::7] = 0xFFFFFFFF                   # EntryJ::setInvalid
    corr["imgIdx_j"][::7] = 0xFFFFFFFF
    keep = ~((corr["imgIdx_i"] == 1) & (corr["imgIdx_j"] == 2))   # pair (1,2) has no matches at all
    a = oracle.solve(campos, normals, intr, corr[keep], pb.poses_init)
    b = oracle.solve(campos, normals, intr, corr[keep][corr[keep]["imgIdx_i"] != 0xFFFFFFFF], pb.poses_init)
    assert np.array_equal(a.poses, b.poses)
    assert np.isfinite(a.poses).all()
    # no correspondences and no dense term: poses only go through Log/Exp
    c = oracle.solve(campos, normals, intr, corr[:0], pb.poses_init, params=oracle.default_params(weight_dense_depth=0.0))
    assert np.abs(c.poses - pb.poses_init).max() < 2e-5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_regression(oracle, path):
    g = np.load(path)
    corr = g["corr"].view(oracle.ENTRYJ_DTYPE).reshape(-1)
    caches = [oracle.build_cache(g["depth"][k], g["normals_full"][k], g["K"], 4.0) for k in range(g["depth"].shape[0])]
    assert np.array_equal(np.stack([c["campos"] for c in caches]), g["campos"])
    prm = oracle.default_params(weight_dense_depth=float(g["weight_dense"]), weight_sparse=float(g["weight_sparse"]), n_threads=1)
    tr = oracle.solve(g["campos"], g["normals"], g["intr"], corr, g["poses_init"], params=prm)
    assert np.array_equal(tr.dense_count, g["dense_count"])
    assert np.allclose(tr.T_after, g["T_after"], atol=1e-6)
    assert np.allclose(tr.poses, g["poses_out"], atol=1e-6)


def test_bench_parity_leg_accepts_the_oracle_and_rejects_a_moved_pose():
    """bench.py's `parity` field: the timed run's output poses of a few instances against the oracle.  Fed the oracle's own result it reports ~0 and ok;
    with one frame rotated by 1e-3 rad it must report that and fail the bar (bench.py then exits non-zero)."""
    import bench
    from bundletrack_amd import synthetic as S
    cfg = dict(K=3, m=40, w_dense=1.0)
    insts = []
    for seed in (3, 4):
        pb = S.make_problem(cfg["K"], cfg["m"], seed, background=False, full_res=False)
        campos, normals, intr = S.analytic_cache(pb)
        insts.append(dict(campos=campos, normals=normals, intr=intr, corr=pb.corr, poses=pb.poses_init))
    from oracle import oracle as O
    ref = [O.solve(q["campos"], q["normals"], q["intr"], q["corr"], q["poses"], params=O.default_params(weight_dense_depth=1.0), want_trace=False).poses for q in insts]
    out = np.stack([np.asarray(r, np.float32).reshape(cfg["K"], 4, 4) for r in ref])
    good = bench.oracle_parity(cfg, insts, [0, 1], out)
    assert good["ok"] and good["worst_rot"] < 1e-6 and good["instances"] == 2
    moved = out.copy()
    c, s = np.cos(1e-3), np.sin(1e-3)
    moved[1, 2, :3, :3] = moved[1, 2, :3, :3] @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
    bad = bench.oracle_parity(cfg, insts, [0, 1], moved)
    assert not bad["ok"] and 5e-4 < bad["worst_rot"] < 2e-3
