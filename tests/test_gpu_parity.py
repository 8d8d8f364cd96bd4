"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.

Tolerance: BASELINE.json's north star -- poses within 1e-4 rad / 1e-4 m of the reference arithmetic PER
Gauss-Newton iterate (fp32 everywhere).  Index/byte work (frame cache, bucketing, match counts) is exact.
"""
import glob
import os

import numpy as np
import pytest

from bundletrack_amd import _lib, synthetic as S

pytestmark = pytest.mark.gpu
TOL_R, TOL_T = 1e-4, 1e-4
GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")) if not os.path.basename(p).startswith("ref_"))


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available()
    from bundletrack_amd.optimizer import BatchSolver, OptimizerGpu, Workspace, build_cache
    ws = Workspace()

    class G:
        pass
    g = G()
    g.torch, g.ws, g.dev = torch, ws, torch.device("cuda:0")
    g.BatchSolver, g.OptimizerGpu, g.build_cache = BatchSolver, OptimizerGpu, build_cache
    # the native extension must be what runs: libbtba.so is loaded in-process
    assert os.path.exists(_lib.LIB_PATH)
    return g


def upload_frames(g, pb):
    d = [g.torch.from_numpy(pb.depth[k]).to(g.dev) for k in range(pb.n_frames)]
    n = [g.torch.from_numpy(pb.normals[k]).to(g.dev) for k in range(pb.n_frames)]
    return d, n


def oracle_cache(oracle, pb):
    caches = [oracle.build_cache(pb.depth[k], pb.normals[k], pb.K, pb.downscale) for k in range(pb.n_frames)]
    return np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches]), caches[0]["intr"], caches


def batch_inputs(g, bs, campos, normals, corr_list, poses_list):
    N = campos.shape[1]
    corr, offs, mx = bs.pack_correspondences(corr_list, N)
    B = len(corr_list)
    corr_d = g.torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(g.dev)
    offs_d = g.torch.from_numpy(offs.astype(np.int32)).to(g.dev)
    poses_d = g.torch.from_numpy(np.stack(poses_list).astype(np.float32)).to(g.dev)
    return corr_d, offs_d, mx, poses_d


def assert_iterates_close(got_T, ref_T, tol_r=TOL_R, tol_t=TOL_T):
    worst = (0.0, 0.0)
    for it in range(ref_T.shape[0]):
        for k in range(ref_T.shape[1]):
            r, t = S.pose_error(got_T[it, k], ref_T[it, k])
            worst = (max(worst[0], r), max(worst[1], t))
            assert r < tol_r and t < tol_t, f"GN iterate {it} frame {k}: rot {r:.3e} trans {t:.3e}"
    return worst


def test_frame_cache_bit_exact(gpu, oracle, small_problem, small_problem_masked):
    for pb in (small_problem, small_problem_masked):
        d, n = upload_frames(gpu, pb)
        campos, nrm, nvalid, intr = gpu.build_cache(gpu.ws, d, n, pb.H, pb.W, pb.K, 4.0)
        gpu.ws.sync()
        ocam, onrm, ointr, caches = oracle_cache(oracle, pb)
        assert np.array_equal(intr.view(np.uint32), ointr.view(np.uint32))
        assert np.array_equal(campos.cpu().numpy().view(np.uint32), ocam.view(np.uint32))
        assert np.array_equal(nrm.cpu().numpy().view(np.uint32), onrm.view(np.uint32))
        assert nvalid.cpu().numpy().tolist() == [c["n_valid"] for c in caches]


def test_frame_cache_other_downscale_and_invalid_depth(gpu, oracle):
    rng = np.random.default_rng(0)
    H, W = 96, 128
    depth = rng.uniform(0.05, 1.5, (2, H, W)).astype(np.float32)          # some below the 0.1 threshold
    depth[:, :10] = 0
    normals = rng.normal(size=(2, H, W, 4)).astype(np.float32); normals[..., 3] = 0
    d = [gpu.torch.from_numpy(depth[k]).to(gpu.dev) for k in range(2)]
    n = [gpu.torch.from_numpy(normals[k]).to(gpu.dev) for k in range(2)]
    for ds in (2.0, 4.0, 8.0):
        campos, nrm, nvalid, intr = gpu.build_cache(gpu.ws, d, n, H, W, S.NOCS_K, ds)
        gpu.ws.sync()
        for k in range(2):
            c = oracle.build_cache(depth[k], normals[k], S.NOCS_K.astype(np.float32), ds)
            assert np.array_equal(campos[k].cpu().numpy().view(np.uint32), c["campos"].view(np.uint32))
            assert np.array_equal(nrm[k].cpu().numpy().view(np.uint32), c["normals"].view(np.uint32))
            assert int(nvalid[k]) == c["n_valid"]


def test_se3_seams(gpu, oracle):
    """convertMatricesToPosesCU / convertPosesToMatricesCU / convertLiePosesToMatricesCU on the device."""
    import ctypes as C
    rng = np.random.default_rng(1)
    n = 64
    x = np.concatenate([rng.normal(size=(n, 3)) * rng.uniform(1e-5, 3.0, (n, 1)) / 1.7, rng.uniform(-1, 1, (n, 3))], 1).astype(np.float32)
    x[0, :3] = 0; x[1, :3] = [5e-5, 0, 0]; x[2, :3] = [5e-4, 1e-4, 0]     # series branches
    x_d = gpu.torch.from_numpy(x).to(gpu.dev)
    T_d = gpu.torch.zeros((n, 16), device=gpu.dev); Ti_d = gpu.torch.zeros((n, 16), device=gpu.dev); x2_d = gpu.torch.zeros((n, 6), device=gpu.dev)
    L = _lib.lib()
    _lib.check(L.btba_poses_to_matrices(gpu.ws.handle, n, x_d.data_ptr(), T_d.data_ptr(), Ti_d.data_ptr()), "p2m")
    _lib.check(L.btba_matrices_to_poses(gpu.ws.handle, n, T_d.data_ptr(), x2_d.data_ptr()), "m2p")
    gpu.ws.sync()
    T, Ti, x2 = T_d.cpu().numpy().reshape(n, 4, 4), Ti_d.cpu().numpy().reshape(n, 4, 4), x2_d.cpu().numpy()
    for k in range(n):
        Tref = oracle.pose_to_matrix(x[k, :3], x[k, 3:])
        assert np.abs(T[k] - Tref).max() < 2e-5          # (1-cos t)/t^2 cancellation amplifies libm ulps (see test_oracle_math)
        assert np.abs(Ti[k] - oracle.mat4_inverse(Tref)).max() < 4e-5
        r, t = oracle.matrix_to_pose(T[k])
        assert np.abs(x2[k, :3] - r).max() < 2e-5 * max(1.0, np.linalg.norm(r)) and np.abs(x2[k, 3:] - t).max() < 3e-5


# Strict 1e-4 cases are scenes whose *own* summation-order noise floor (oracle with sequential fp32 sums vs
# the canonical exact-sum oracle, i.e. the spread the reference's float atomics can produce run to run) is
# below 1e-5; the "noisy" cases further down are the weakly conditioned 100 %-valid background scenes.
@pytest.mark.parametrize("case", [
    dict(K=3, m=120, seed=21, bg=False, wd=1.0, ws=1.0),
    dict(K=5, m=300, seed=28, bg=False, wd=1.0, ws=1.0),
    dict(K=4, m=200, seed=23, bg=False, wd=1.0, ws=1.0),      # realistic mask: ~5 % valid pixels
    dict(K=6, m=150, seed=24, bg=True, wd=0.0, ws=1.0),       # sparse only (BASELINE config 2 shape)
    dict(K=4, m=0, seed=25, bg=True, wd=1.0, ws=1.0, tol=5e-4),   # no feature matches at all: dense only, 100 % valid -- the class the
                                                                # reference itself determines only to ~1e-3; measured 3e-6 ... 2e-5
    dict(K=2, m=500, seed=26, bg=True, wd=1.0, ws=1.0),
], ids=lambda c: f"K{c['K']}_m{c['m']}_{'bg' if c['bg'] else 'mask'}_wd{c['wd']:g}")
def test_parity_per_gn_iterate(gpu, oracle, case):
    pb = S.make_problem(case["K"], case["m"], case["seed"], background=case["bg"])
    ocam, onrm, ointr, _ = oracle_cache(oracle, pb)
    ref = oracle.solve(ocam, onrm, ointr, pb.corr, pb.poses_init, params=oracle.default_params(weight_dense_depth=case["wd"], weight_sparse=case["ws"]))
    bs = gpu.BatchSolver(gpu.ws, weight_dense_depth=case["wd"], weight_sparse=case["ws"])
    cam_d, nrm_d = gpu.torch.from_numpy(ocam[None]).to(gpu.dev), gpu.torch.from_numpy(onrm[None]).to(gpu.dev)
    corr_d, offs_d, mx, poses_d = batch_inputs(gpu, bs, ocam[None], onrm[None], [pb.corr], [pb.poses_init])
    tr = bs.solve(cam_d, nrm_d, ointr, corr_d, offs_d, mx, poses_d, trace=True)
    tv = bs.trace_view(tr)
    tol = case.get("tol", TOL_R)
    worst = assert_iterates_close(tv.T_after[0], ref.T_after, tol, tol)
    print(f"parity {case}: worst per-iterate diff rot {worst[0]:.2e} trans {worst[1]:.2e}")
    if case["wd"] > 0:
        P = case["K"] * (case["K"] - 1) // 2
        cnt = tv.dense_pair[0, :, :, 27].astype(np.int64)
        # accept/reject ties at the hard thresholds (0.02 m, cos 45 deg, rounded in-bounds test) flip for a handful
        # of the 19 200 pixels per pair once iterates differ in the last bits
        assert np.abs(cnt - ref.dense_count[:, :P]).max() <= 6
    # first linearisation (identical inputs): rhs / preconditioner / first PCG scalars agree tightly
    assert np.abs(tv.rhs[0, 0] - ref.rhs[0]).max() <= 2e-4 * max(1e-6, np.abs(ref.rhs[0]).max())
    assert np.abs(tv.precond[0, 0] - ref.precond[0]).max() <= 1e-4 * np.abs(ref.precond[0]).max()
    if ref.pcg_scalars[0, 0, 0] > 0:
        assert abs(tv.pcg_scalars[0, 0, 0, 1] - ref.pcg_scalars[0, 0, 1]) <= 1e-3 * abs(ref.pcg_scalars[0, 0, 1])
    fin = poses_d.cpu().numpy()[0]
    assert np.array_equal(fin, tv.T_after[0, -1])                     # output = Exp(x) of the last iterate


@pytest.mark.parametrize("case", [dict(K=5, m=300, seed=22), dict(K=3, m=120, seed=21), dict(K=4, m=150, seed=3)],
                         ids=lambda c: f"K{c['K']}_m{c['m']}_bg")
def test_parity_on_weakly_conditioned_scenes(gpu, oracle, case):
    """100 %-valid background-sphere scenes with few frames: five PCG steps on a stiff-translation / soft-rotation
    system amplify last-bit noise ~1000x, so the reference itself (float atomics, arbitrary order) is not
    reproducible to 1e-4 here (its own summation-order spread, measured below with the oracle, reaches 4e-4).
    These scenes are therefore held to 1e-3, not to the 1e-4 parity bar; the spread is printed for the record."""
    pb = S.make_problem(case["K"], case["m"], case["seed"], background=True)
    ocam, onrm, ointr, _ = oracle_cache(oracle, pb)
    ref = oracle.solve(ocam, onrm, ointr, pb.corr, pb.poses_init)
    seq = oracle.solve(ocam, onrm, ointr, pb.corr, pb.poses_init, params=oracle.default_params(accum_mode=0))
    floor = max(max(S.pose_error(ref.T_after[it, k], seq.T_after[it, k])) for it in range(7) for k in range(case["K"]))
    tol = 1e-3
    bs = gpu.BatchSolver(gpu.ws)
    cam_d, nrm_d = gpu.torch.from_numpy(ocam[None]).to(gpu.dev), gpu.torch.from_numpy(onrm[None]).to(gpu.dev)
    corr_d, offs_d, mx, poses_d = batch_inputs(gpu, bs, ocam[None], onrm[None], [pb.corr], [pb.poses_init])
    tv = bs.trace_view(bs.solve(cam_d, nrm_d, ointr, corr_d, offs_d, mx, poses_d, trace=True))
    # decisions first, then values: an iterate may exceed the 1e-4 bar only from the first differing accept / guard decision on
    from helpers import check_parity_with_decisions, first_decision_divergence
    div = first_decision_divergence(tv.pcg_scalars[0], tv.dense_pair[0][..., 27], ref.pcg_scalars, ref.dense_count)
    # (round 6) the floor under an iterate before the first differing decision: the oracle's summation spread AND -- where oracle/_ref is there -- what the reference's own
    # code does to itself on this scene (helpers.reference_licence); the former alone moves with the box's OpenMP thread count
    cum = None
    from oracle import reference as R
    if os.path.exists(R.SO_SOLVER) and os.path.exists(R.SO_SOLVER_FM):
        from helpers import reference_licence
        cum, _ = reference_licence(R, S.pose_error, ocam, onrm, ointr, pb.corr, pb.poses_init)
    worst = check_parity_with_decisions(tv.T_after[0], ref.T_after, div, S.pose_error, 1e-4, tol, f"weakly conditioned {case}", spread_T=seq.T_after, ref_spread=cum)
    print(f"first differing decision: {div}")
    print(f"weakly conditioned {case}: oracle summation spread {floor:.2e}, HIP vs oracle worst {max(worst):.2e}, bar {tol:.2e}")


def test_assembled_matrix_equals_reference_operator(gpu, oracle, small_problem):
    """A (sparse part built from moment sums + dense S blocks) applied to a vector equals the reference's
    matrix-free applyJ/applyJT plus its dense JtJ mat-vec (SolverBundlingEquationsLie.h:140-211,
    SolverBundlingDenseUtil.h:349-385)."""
    pb = small_problem
    N = pb.n_frames
    ocam, onrm, ointr, _ = oracle_cache(oracle, pb)
    ref = oracle.solve(ocam, onrm, ointr, pb.corr, pb.poses_init, params=oracle.default_params(n_gn_iters=1))
    bs = gpu.BatchSolver(gpu.ws, n_gn_iters=1)
    cam_d, nrm_d = gpu.torch.from_numpy(ocam[None]).to(gpu.dev), gpu.torch.from_numpy(onrm[None]).to(gpu.dev)
    corr_d, offs_d, mx, poses_d = batch_inputs(gpu, bs, ocam[None], onrm[None], [pb.corr], [pb.poses_init])
    tv = bs.trace_view(bs.solve(cam_d, nrm_d, ointr, corr_d, offs_d, mx, poses_d, trace=True))
    A = tv.A[0, 0].astype(np.float64)
    assert np.abs(A - A.T).max() <= 1e-6 * np.abs(A).max()
    assert np.all(A[:6] == 0) and np.all(A[:, :6] == 0)               # frame 0 is fixed
    rng = np.random.default_rng(3)
    p = rng.normal(size=(N, 6)).astype(np.float32); p[0] = 0           # (rot, trans)
    # T at the linearisation point = Exp(Log(poses_init)) like the reference
    T0 = np.stack([oracle.pose_to_matrix(*oracle.matrix_to_pose(pb.poses_init[k])) for k in range(N)])
    sp = oracle.sparse_apply(pb.corr, T0, p).astype(np.float64)        # (rot, trans)
    pv = np.concatenate([p[:, 3:], p[:, :3]], 1).reshape(-1).astype(np.float64)   # [trans, rot]
    dense = (ref.dense_JtJ[0].astype(np.float64) @ pv).reshape(N, 6)
    want = np.concatenate([sp[:, 3:], sp[:, :3]], 1) + dense
    got = (A @ pv).reshape(N, 6)
    assert np.abs(got[1:] - want[1:]).max() <= 3e-4 * np.abs(want).max()


def test_drop_in_boundary_matches_oracle(gpu, oracle, small_problem_masked):
    """OptimizerGpu::optimizeFrames: full-resolution device frames + host EntryJ + host poses in/out."""
    pb = small_problem_masked
    d, n = upload_frames(gpu, pb)
    ocam, onrm, ointr, _ = oracle_cache(oracle, pb)
    ref = oracle.solve(ocam, onrm, ointr, pb.corr, pb.poses_init)
    poses = pb.poses_init.copy()
    opt = gpu.OptimizerGpu({"bundle": {"num_iter_outter": 7, "num_iter_inner": 5, "robust_delta": 0.005, "image_downscale": 4},
                            "p2p": {"max_dist": 0.02, "max_normal_angle": 45}})
    out = opt.optimizeFrames(pb.corr, pb.n_match_per_pair, pb.n_frames, pb.H, pb.W, d, None, n, poses, pb.K)
    assert out is poses                                             # in/out like the reference's `poses&`
    for k in range(pb.n_frames):
        r, t = S.pose_error(poses[k], ref.poses[k])
        assert r < TOL_R and t < TOL_T
    st = opt.last_stats
    assert st["n_corr"] == len(pb.corr) and st["n_frames"] == pb.n_frames and st["ms_solve"] > 0
    # stateless variant (ws = NULL: allocate and free inside the call, like the reference) gives identical bits
    poses2 = pb.poses_init.copy()
    gpu.OptimizerGpu(workspace=None).optimizeFrames(pb.corr, None, pb.n_frames, pb.H, pb.W, d, None, n, poses2, pb.K)
    assert np.array_equal(poses, poses2)
    # shuffled (non pair-major) correspondences with invalid entries sprinkled in: same optimum within tolerance
    rng = np.random.default_rng(5)
    corr = np.concatenate([pb.corr, pb.corr[:37]])
    corr["imgIdx_i"][-37:] = 0xFFFFFFFF
    corr = corr[rng.permutation(len(corr))]
    poses3 = pb.poses_init.copy()
    opt.optimizeFrames(corr, None, pb.n_frames, pb.H, pb.W, d, None, n, poses3, pb.K)
    for k in range(pb.n_frames):
        r, t = S.pose_error(poses3[k], ref.poses[k])
        assert r < TOL_R and t < TOL_T


def test_error_codes_not_exits(gpu, small_problem):
    pb = small_problem
    d, n = upload_frames(gpu, pb)
    opt = gpu.OptimizerGpu(workspace=gpu.ws)
    with pytest.raises(_lib.BtbaError) as e:                         # n_frames < 2: MLIB_ASSERT in the reference
        opt.optimizeFrames(pb.corr[:0], None, 1, pb.H, pb.W, d[:1], None, n[:1], pb.poses_init[:1].copy(), pb.K)
    assert e.value.status == _lib.BTBA_EINVAL
    bad = pb.corr.copy(); bad["imgIdx_j"][3] = 99
    with pytest.raises(_lib.BtbaError) as e:
        opt.optimizeFrames(bad, None, pb.n_frames, pb.H, pb.W, d, None, n, pb.poses_init.copy(), pb.K)
    assert e.value.status == _lib.BTBA_EINVAL
    poses = pb.poses_init.copy(); poses[1, 0, 0] = np.nan
    with pytest.raises(_lib.BtbaError) as e:
        opt.optimizeFrames(pb.corr, None, pb.n_frames, pb.H, pb.W, d, None, n, poses, pb.K)
    assert e.value.status == _lib.BTBA_ENUMERIC
    assert np.isnan(poses[1, 0, 0])                                  # caller's buffer untouched on failure


def test_pair_policies(gpu, oracle, small_problem_masked):
    """TARGET_MORE_VALID / EXPLICIT orientation incl. the literal FlipJtJ erasure when target > source."""
    pb = small_problem_masked
    N = pb.n_frames
    d, n = upload_frames(gpu, pb)
    ocam, onrm, ointr, caches = oracle_cache(oracle, pb)
    nv = [c["n_valid"] for c in caches]
    pairs = np.array([(i, j) if nv[i] >= nv[j] else (j, i) for i in range(N) for j in range(i + 1, N)], np.int32)
    ref = oracle.solve(ocam, onrm, ointr, pb.corr, pb.poses_init, pairs=pairs)
    for policy, kw in ((_lib.PAIRS_TARGET_MORE_VALID, {}), (_lib.PAIRS_EXPLICIT, dict(dense_pairs=pairs))):
        poses = pb.poses_init.copy()
        opt = gpu.OptimizerGpu(workspace=gpu.ws, pair_policy=policy)
        opt.optimizeFrames(pb.corr, None, N, pb.H, pb.W, d, None, n, poses, pb.K, **kw)
        for k in range(N):
            r, t = S.pose_error(poses[k], ref.poses[k])
            assert r < TOL_R and t < TOL_T
    # a reversed list (every target > source) must lose its cross blocks, exactly like the oracle's flip
    rev = np.array([(j, i) for i in range(N) for j in range(i + 1, N)], np.int32)
    ref2 = oracle.solve(ocam, onrm, ointr, pb.corr, pb.poses_init, pairs=rev)
    poses = pb.poses_init.copy()
    gpu.OptimizerGpu(workspace=gpu.ws, pair_policy=_lib.PAIRS_EXPLICIT).optimizeFrames(pb.corr, None, N, pb.H, pb.W, d, None, n, poses, pb.K, dense_pairs=rev)
    for k in range(N):
        r, t = S.pose_error(poses[k], ref2.poses[k])
        assert r < TOL_R and t < TOL_T


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_fixtures(gpu, path):
    """Committed fixtures (self-derived, see tests/golden/make_golden.py): no oracle call at run time."""
    gd = np.load(path)
    corr = gd["corr"].view(_lib.ENTRYJ_DTYPE).reshape(-1)
    bs = gpu.BatchSolver(gpu.ws, weight_dense_depth=float(gd["weight_dense"]), weight_sparse=float(gd["weight_sparse"]))
    cam_d, nrm_d = gpu.torch.from_numpy(gd["campos"][None]).to(gpu.dev), gpu.torch.from_numpy(gd["normals"][None]).to(gpu.dev)
    corr_d, offs_d, mx, poses_d = batch_inputs(gpu, bs, gd["campos"][None], gd["normals"][None], [corr], [gd["poses_init"]])
    tv = bs.trace_view(bs.solve(cam_d, nrm_d, gd["intr"], corr_d, offs_d, mx, poses_d, trace=True))
    assert_iterates_close(tv.T_after[0], gd["T_after"])
    # full boundary on the stored full-resolution frames
    N = gd["depth"].shape[0]
    d = [gpu.torch.from_numpy(gd["depth"][k]).to(gpu.dev) for k in range(N)]
    n = [gpu.torch.from_numpy(gd["normals_full"][k]).to(gpu.dev) for k in range(N)]
    poses = gd["poses_init"].copy()
    gpu.OptimizerGpu(workspace=gpu.ws, weight_dense_depth=float(gd["weight_dense"]), weight_sparse=float(gd["weight_sparse"])).optimizeFrames(
        corr, gd["n_match_per_pair"], N, gd["depth"].shape[1], gd["depth"].shape[2], d, None, n, poses, gd["K"])
    for k in range(N):
        r, t = S.pose_error(poses[k], gd["poses_out"][k])
        assert r < TOL_R and t < TOL_T


def test_batch_equals_single_and_is_deterministic(gpu, oracle):
    """Instances in one grid do not interact; repeated runs are bit-identical (no float atomics)."""
    pbs = [S.make_problem(4, 180 + 20 * b, seed=40 + b, background=False) for b in range(5)]   # ragged corr counts, well-conditioned scenes
    cams, nrms, intr = [], [], None
    for pb in pbs:
        c, n_, intr, _ = oracle_cache(oracle, pb)
        cams.append(c); nrms.append(n_)
    bs = gpu.BatchSolver(gpu.ws)
    cam_d, nrm_d = gpu.torch.from_numpy(np.stack(cams)).to(gpu.dev), gpu.torch.from_numpy(np.stack(nrms)).to(gpu.dev)
    corr_d, offs_d, mx, poses_d = batch_inputs(gpu, bs, None if False else np.stack(cams), None, [pb.corr for pb in pbs], [pb.poses_init for pb in pbs])
    p0 = poses_d.clone()
    bs.solve(cam_d, nrm_d, intr, corr_d, offs_d, mx, poses_d)
    gpu.ws.sync()
    batch = poses_d.cpu().numpy()
    poses_d2 = p0.clone()
    bs.solve(cam_d, nrm_d, intr, corr_d, offs_d, mx, poses_d2)
    gpu.ws.sync()
    assert np.array_equal(batch, poses_d2.cpu().numpy())
    for b, pb in enumerate(pbs):
        ref = oracle.solve(cams[b], nrms[b], intr, pb.corr, pb.poses_init)
        for k in range(4):
            r, t = S.pose_error(batch[b, k], ref.poses[k])
            assert r < TOL_R and t < TOL_T, (b, k, r, t)
    # single-instance launch of instance 3 uses different tile/chunk counts -> same result within fp32 noise
    one = p0[3:4].clone()
    bs.solve(cam_d[3:4], nrm_d[3:4], intr, corr_d[3:4], offs_d[3:4], mx, one)
    gpu.ws.sync()
    for k in range(4):
        r, t = S.pose_error(one.cpu().numpy()[0, k], batch[3, k])
        assert r < 2e-5 and t < 2e-5


def test_correspondence_order_invariance(gpu, oracle, small_problem_masked):
    """Property: shuffling correspondences inside their pair segments changes nothing beyond fp32 noise.
    (Masked scene: the K=4 background-sphere scene amplifies last-bit noise ~1000x through the 5-step PCG --
    see test_oracle_solver.test_thread_count_and_accum_mode -- and would need a 5e-4 tolerance.)"""
    pb = small_problem_masked
    ocam, onrm, ointr, _ = oracle_cache(oracle, pb)
    bs = gpu.BatchSolver(gpu.ws)
    cam_d, nrm_d = gpu.torch.from_numpy(ocam[None]).to(gpu.dev), gpu.torch.from_numpy(onrm[None]).to(gpu.dev)
    rng = np.random.default_rng(7)
    outs = []
    for trial in range(3):
        corr = pb.corr.copy()
        if trial:
            o = 0
            for m in pb.n_match_per_pair:
                corr[o:o + m] = corr[o:o + m][rng.permutation(m)]
                o += m
        corr_d, offs_d, mx, poses_d = batch_inputs(gpu, bs, ocam[None], onrm[None], [corr], [pb.poses_init])
        bs.solve(cam_d, nrm_d, ointr, corr_d, offs_d, mx, poses_d)
        gpu.ws.sync()
        outs.append(poses_d.cpu().numpy()[0])
    for o in outs[1:]:
        for k in range(pb.n_frames):
            r, t = S.pose_error(o[k], outs[0][k])
            assert r < 2e-5 and t < 2e-5


def test_tile_and_chunk_counts_do_not_change_the_result(gpu, oracle, small_problem_masked):
    pb = small_problem_masked
    ocam, onrm, ointr, _ = oracle_cache(oracle, pb)
    cam_d, nrm_d = gpu.torch.from_numpy(ocam[None]).to(gpu.dev), gpu.torch.from_numpy(onrm[None]).to(gpu.dev)
    outs = []
    for tiles, chunks in ((1, 1), (3, 2), (25, 1), (75, 4)):
        bs = gpu.BatchSolver(gpu.ws, dense_tiles=tiles, sparse_chunks=chunks)
        corr_d, offs_d, mx, poses_d = batch_inputs(gpu, bs, ocam[None], onrm[None], [pb.corr], [pb.poses_init])
        bs.solve(cam_d, nrm_d, ointr, corr_d, offs_d, mx, poses_d)
        gpu.ws.sync()
        outs.append(poses_d.cpu().numpy()[0])
    for o in outs[1:]:
        for k in range(pb.n_frames):
            r, t = S.pose_error(o[k], outs[0][k])
            assert r < 2e-5 and t < 2e-5


def test_compact_cache_matches_float4_cache(gpu, oracle, small_problem_masked):
    """The compact (z, nx, ny, nz) cache re-derives camPos with the cache builder's exact fp32 operations (checked
    bit for bit below); the two sweep kernels are separate compilations whose optional FMA contraction differs, so
    the solver outputs agree to fp32 round-off (1e-7 relative on the pair sums), not bit for bit."""
    from bundletrack_amd.optimizer import build_cache_zn, pack_zn
    for pb in (small_problem_masked, S.make_problem(5, 250, seed=62, background=False)):
        d, n = upload_frames(gpu, pb)
        campos, nrm, nvalid, intr = gpu.build_cache(gpu.ws, d, n, pb.H, pb.W, pb.K, 4.0)
        zn, nvalid2, intr2 = build_cache_zn(gpu.ws, d, n, pb.H, pb.W, pb.K, 4.0)
        gpu.ws.sync()
        assert np.array_equal(intr, intr2) and np.array_equal(nvalid.cpu().numpy(), nvalid2.cpu().numpy())
        znh, camh, nrmh = zn.cpu().numpy(), campos.cpu().numpy(), nrm.cpu().numpy()
        assert np.array_equal(znh[..., 1:], nrmh[..., :3])
        valid = camh[..., 3] == 1
        assert np.array_equal(znh[..., 0][valid], camh[..., 2][valid])                 # z = depth where valid
        assert np.array_equal(pack_zn(gpu.ws, campos, nrm).cpu().numpy()[valid], znh[valid])
        bs = gpu.BatchSolver(gpu.ws)
        corr_d, offs_d, mx, poses_a = batch_inputs(gpu, bs, camh[None], None, [pb.corr], [pb.poses_init])
        poses_b = poses_a.clone()
        ta = bs.trace_view(bs.solve(campos[None], nrm[None], intr, corr_d, offs_d, mx, poses_a, trace=True))
        tb = bs.trace_view(bs.solve_zn(zn[None], pb.H, pb.W, pb.K, corr_d, offs_d, mx, poses_b, trace=True))
        # camPos re-derived on the host with the same fp32 operations == the float4 cache, bit for bit
        K4 = np.eye(4, dtype=np.float32); K4[:3, :3] = pb.K
        Ki = oracle.mat4_inverse(K4).reshape(16)
        xi, yi = S.cache_source_pixels(pb.H, pb.W, camh.shape[1], camh.shape[2])
        dd = znh[..., 0]
        vx = xi[None, None, :].astype(np.float32) * dd; vy = yi[None, :, None].astype(np.float32) * dd
        assert np.array_equal((Ki[0] * vx + Ki[2] * dd)[valid], camh[..., 0][valid]) and np.array_equal((Ki[5] * vy + Ki[6] * dd)[valid], camh[..., 1][valid])
        assert np.array_equal(ta.dense_pair[:, 0, :, 27], tb.dense_pair[:, 0, :, 27])          # same accepted pixels at identical poses
        # later iterates differ at round-off, which may flip a pixel sitting on an acceptance threshold
        assert np.abs(ta.dense_pair[..., 27] - tb.dense_pair[..., 27]).max() <= 2
        assert np.abs(ta.dense_pair[:, 0] - tb.dense_pair[:, 0]).max() <= 2e-6 * np.abs(ta.dense_pair[:, 0]).max()    # first linearisation: round-off only
        pa, pbb = poses_a.cpu().numpy()[0], poses_b.cpu().numpy()[0]
        for k in range(pb.n_frames):
            r, t = S.pose_error(pa[k], pbb[k])
            assert r < 2e-5 and t < 2e-5
        p1, p2 = pb.poses_init.copy(), pb.poses_init.copy()
        gpu.OptimizerGpu(workspace=gpu.ws).optimizeFrames(pb.corr, None, pb.n_frames, pb.H, pb.W, d, None, n, p1, pb.K)
        gpu.OptimizerGpu(workspace=gpu.ws, flags=_lib.FLAG_FLOAT4_CACHE).optimizeFrames(pb.corr, None, pb.n_frames, pb.H, pb.W, d, None, n, p2, pb.K)
        for k in range(pb.n_frames):
            r, t = S.pose_error(p1[k], p2[k])
            assert r < 2e-5 and t < 2e-5


def test_compact_cache_general_intrinsics(gpu, oracle):
    """Skewed K (non-zero K[0,1]): the general back-projection path, against the oracle."""
    K = S.NOCS_K.copy(); K[0, 1] = 3.7
    pb = S.make_problem(3, 200, seed=61, background=False, K=K)
    d, n = upload_frames(gpu, pb)
    ocam, onrm, ointr, _ = oracle_cache(oracle, pb)
    ref = oracle.solve(ocam, onrm, ointr, pb.corr, pb.poses_init)
    poses = pb.poses_init.copy()
    gpu.OptimizerGpu(workspace=gpu.ws).optimizeFrames(pb.corr, None, pb.n_frames, pb.H, pb.W, d, None, n, poses, pb.K)
    for k in range(pb.n_frames):
        r, t = S.pose_error(poses[k], ref.poses[k])
        assert r < TOL_R and t < TOL_T


def test_valid_pixel_compaction(gpu, oracle, small_problem, small_problem_masked):
    """Walking each source frame's list of pixels with a depth (BTBA_FLAG_COMPACTION; chosen automatically by
    optimize_frames for masked frames) vs all Wd x Hd pixels: round-off-level agreement (different kernel
    instantiation / lane assignment); the accepted-pixel counts are identical either way."""
    from bundletrack_amd.optimizer import build_cache_zn
    for pb, full in ((small_problem, True), (small_problem_masked, False)):
        d, n = upload_frames(gpu, pb)
        zn, nvalid, intr = build_cache_zn(gpu.ws, d, n, pb.H, pb.W, pb.K, 4.0)
        outs, cnts = [], []
        for flags in (_lib.FLAG_COMPACTION, 0):
            bs = gpu.BatchSolver(gpu.ws, flags=flags)
            corr_d, offs_d, mx, poses_d = batch_inputs(gpu, bs, zn.cpu().numpy()[None], None, [pb.corr], [pb.poses_init])
            tv = bs.trace_view(bs.solve_zn(zn[None], pb.H, pb.W, pb.K, corr_d, offs_d, mx, poses_d, trace=True))
            outs.append(poses_d.cpu().numpy()[0]); cnts.append(tv.dense_pair[0, :, :, 27].copy())
        if full:
            assert int(nvalid.min()) == zn.shape[1] * zn.shape[2]
            assert np.array_equal(cnts[0][0], cnts[1][0])
            for k in range(pb.n_frames):
                r, t = S.pose_error(outs[0][k], outs[1][k])
                assert r < 1e-3 and t < 1e-3            # weakly conditioned 100 %-valid K=4 scene, see test_parity_on_weakly_conditioned_scenes
        else:
            assert int(nvalid.max()) < 0.2 * zn.shape[1] * zn.shape[2]
            assert np.array_equal(cnts[0][0], cnts[1][0])            # first linearisation: same accepted pixels
            for k in range(pb.n_frames):
                r, t = S.pose_error(outs[0][k], outs[1][k])
                assert r < 2e-5 and t < 2e-5


def sub_problem(pb, frames):
    """The BA window made of `frames` (indices into pb, ascending): correspondences re-indexed pair-major."""
    idx = {f: k for k, f in enumerate(frames)}
    keep = np.isin(pb.corr["imgIdx_i"], frames) & np.isin(pb.corr["imgIdx_j"], frames)
    corr = pb.corr[keep].copy()
    corr["imgIdx_i"] = [idx[i] for i in corr["imgIdx_i"]]
    corr["imgIdx_j"] = [idx[j] for j in corr["imgIdx_j"]]
    return corr, pb.poses_init[frames].copy()


@pytest.mark.parametrize("background", [False, True], ids=["masked", "full"])
def test_persistent_frame_cache(gpu, background):
    """btba_optimize_frames_keyed: frames cached by earlier calls are not cached again and the
    results are bit-identical to the stateless entry point, for sliding windows, recycled slots and changed buffers."""
    from bundletrack_amd.optimizer import Workspace, frame_cache_clear
    pb = S.make_problem(7, 120, seed=71, background=background)
    d, n = upload_frames(gpu, pb)
    ws = Workspace()
    keyed, plain = gpu.OptimizerGpu(workspace=ws), gpu.OptimizerGpu(workspace=gpu.ws)
    windows = [[0, 1, 2, 3], [0, 1, 2, 3], [1, 2, 3, 4], [0, 2, 4, 5], [3, 4, 5, 6], [0, 1, 2, 3]]
    built = []
    for w in windows:
        corr, p0 = sub_problem(pb, w)
        pa, pk = p0.copy(), p0.copy()
        plain.optimizeFrames(corr, None, len(w), pb.H, pb.W, [d[f] for f in w], None, [n[f] for f in w], pa, pb.K)
        keyed.optimizeFrames(corr, None, len(w), pb.H, pb.W, [d[f] for f in w], None, [n[f] for f in w], pk, pb.K, frame_keys=[100 + f for f in w])
        assert np.array_equal(pa, pk), w
        built.append(keyed.last_stats["cache_frames_built"])
        assert plain.last_stats["cache_frames_built"] == len(w)
    assert built == [4, 0, 1, 1, 1, 0]                       # only frames never seen before are cached
    # the same key on a different device buffer is a different frame: rebuilt, and the result follows the new data
    w = [0, 1, 2, 3]
    corr, p0 = sub_problem(pb, w)
    d_alt = [d[f] for f in w]; n_alt = [n[f] for f in w]
    d_alt[2] = d[2].clone(); n_alt[2] = n[2].clone()
    pk = p0.copy()
    keyed.optimizeFrames(corr, None, 4, pb.H, pb.W, d_alt, None, n_alt, pk, pb.K, frame_keys=[100 + f for f in w])
    assert keyed.last_stats["cache_frames_built"] == 1
    # more distinct frames than pool slots (32): least-recently-used slots are recycled, results stay exact
    for rep in range(12):
        keys = [1000 + 4 * rep + k for k in range(4)]
        pk2 = p0.copy()
        keyed.optimizeFrames(corr, None, 4, pb.H, pb.W, [d[f] for f in w], None, [n[f] for f in w], pk2, pb.K, frame_keys=keys)
        assert keyed.last_stats["cache_frames_built"] == 4 and np.array_equal(pk2, pk)
    # clear: everything is rebuilt
    frame_cache_clear(ws)
    keyed.optimizeFrames(corr, None, 4, pb.H, pb.W, [d[f] for f in w], None, [n[f] for f in w], p0.copy(), pb.K, frame_keys=[100 + f for f in w])
    assert keyed.last_stats["cache_frames_built"] == 4
    # argument errors: duplicate keys, reference-layout cache, no workspace
    with pytest.raises(_lib.BtbaError):
        keyed.optimizeFrames(corr, None, 4, pb.H, pb.W, [d[f] for f in w], None, [n[f] for f in w], p0.copy(), pb.K, frame_keys=[1, 2, 2, 3])
    f4 = gpu.OptimizerGpu(workspace=ws)
    f4.params.flags |= _lib.FLAG_FLOAT4_CACHE
    with pytest.raises(_lib.BtbaError):
        f4.optimizeFrames(corr, None, 4, pb.H, pb.W, [d[f] for f in w], None, [n[f] for f in w], p0.copy(), pb.K, frame_keys=[1, 2, 3, 4])
    with pytest.raises(ValueError):
        gpu.OptimizerGpu(workspace=None).optimizeFrames(corr, None, 4, pb.H, pb.W, [d[f] for f in w], None, [n[f] for f in w], p0.copy(), pb.K, frame_keys=[1, 2, 3, 4])


def test_edge_shapes_through_the_boundary(gpu, oracle):
    """Sizes at the edges of what the boundary accepts, each against the oracle: the largest window (N = 31, the 186
    unknowns whose normal matrix fits one CU's LDS), a frame size that is not a multiple of the downscale, ragged correspondence
    segments (empty pairs next to full ones), frames without a single valid pixel, and N = 41 rejected with a status."""
    opt = gpu.OptimizerGpu(workspace=gpu.ws)

    def run(pb, corr=None, depth=None, normals=None):
        corr = pb.corr if corr is None else corr
        depth = pb.depth if depth is None else depth
        normals = pb.normals if normals is None else normals
        d = [gpu.torch.from_numpy(depth[k]).to(gpu.dev) for k in range(pb.n_frames)]
        n = [gpu.torch.from_numpy(normals[k]).to(gpu.dev) for k in range(pb.n_frames)]
        caches = [oracle.build_cache(depth[k], normals[k], pb.K, pb.downscale) for k in range(pb.n_frames)]
        ref = oracle.solve(np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches]), caches[0]["intr"], corr, pb.poses_init)
        poses = pb.poses_init.copy()
        opt.optimizeFrames(corr, None, pb.n_frames, pb.H, pb.W, d, None, n, poses, pb.K)
        worst = max(max(S.pose_error(poses[k], ref.poses[k])) for k in range(pb.n_frames))
        assert np.isfinite(poses).all()
        return worst, poses, d, n

    # N = 31 (BTBA_MAX_FRAMES_LDS: the largest window whose matrix lives in LDS) on small frames (128 x 96 -> 32 x 24 cache;
    # intrinsics scaled with the image)
    Ks = S.NOCS_K.copy(); Ks[:2] *= 0.2
    big = S.make_problem(31, 12, seed=91, background=False, H=96, W=128, K=Ks, rot_step_deg=(5.0, 6.0))
    worst, _, dN, nN = run(big)
    print(f"N=31: worst pose diff {worst:.2e}")
    assert worst < 5e-4, worst                           # 465 pairs of 12 matches each: dense-dominated, measured 8e-7 ... 4e-5
    # N = 32, 40 and 85 (= BTBA_MAX_FRAMES = the reference's MAX_NUM_IMAGES): the matrix moves to the global scratch and the
    # PCG to 16 waves; same arithmetic, same bar
    for N_big, seed in ((32, 93), (40, 94), (85, 95)):
        pbN = S.make_problem(N_big, 12, seed=seed, background=False, H=96, W=128, K=Ks, rot_step_deg=(5.0, 6.0))
        worst, posesN, dB, nB = run(pbN)
        print(f"N={N_big}: worst pose diff {worst:.2e}")
        assert worst < 5e-4, (N_big, worst)
        if N_big == 40:
            # the same window with reduction and assembly on ONE workgroup (the path a traced solve takes) instead of k_big_reduce /
            # k_big_assemble: the same sums in another grouping
            gpu.ws.set_option(_lib.OPT_BIG_ASSEMBLY, 0)
            try:
                _, poses_one, _, _ = run(pbN)
            finally:
                gpu.ws.set_option(_lib.OPT_BIG_ASSEMBLY, 1)
            d_paths = max(max(S.pose_error(posesN[k], poses_one[k])) for k in range(N_big))
            print(f"N=40: many-workgroup assembly vs one workgroup: worst pose diff {d_paths:.2e}")
            assert 0 < d_paths < 5e-5 or np.array_equal(posesN, poses_one), d_paths
    with pytest.raises(_lib.BtbaError) as e:             # N = 86 is refused with a status, not `while(1);` (SolverBundling.cu:621-625)
        opt.optimizeFrames(pbN.corr[:0], None, 86, pbN.H, pbN.W, dB + dB[:1], None, nB + nB[:1], np.tile(np.eye(4, dtype=np.float32), (86, 1, 1)), pbN.K)
    assert e.value.status == _lib.BTBA_EINVAL
    # 53 x 37 frames: int(W / 4) = 13, int(H / 4) = 9 (LossGPU.cu:56-57), nearest-neighbour resample with fractional scales
    Ko = S.NOCS_K.copy(); Ko[0] *= 53 / 640; Ko[1] *= 37 / 480
    odd = S.make_problem(4, 150, seed=92, background=True, H=37, W=53, K=Ko)
    worst, _, _, _ = run(odd)
    assert worst < TOL_R, worst
    # ragged segments: two pairs lose every correspondence, one keeps a single one
    pb = S.make_problem(5, 200, seed=93, background=False)
    keep = np.ones(len(pb.corr), bool)
    i, j = pb.corr["imgIdx_i"], pb.corr["imgIdx_j"]
    keep[(i == 0) & (j == 1)] = False
    keep[(i == 2) & (j == 3)] = False
    one = np.nonzero((i == 1) & (j == 4))[0]
    keep[one[1:]] = False
    worst, _, _, _ = run(pb, corr=pb.corr[keep])
    assert worst < TOL_R, worst
    # a window whose frames carry no depth at all: the dense term vanishes, the feature term alone drives the solve
    zd, zn = np.zeros_like(pb.depth), np.zeros_like(pb.normals)
    worst, poses_nodepth, _, _ = run(pb, depth=zd, normals=zn)
    assert worst < TOL_R, worst
    sparse_only = gpu.OptimizerGpu(workspace=gpu.ws)
    sparse_only.params.weight_dense_depth = 0.0
    p2 = pb.poses_init.copy()
    d = [gpu.torch.from_numpy(pb.depth[k]).to(gpu.dev) for k in range(pb.n_frames)]
    n = [gpu.torch.from_numpy(pb.normals[k]).to(gpu.dev) for k in range(pb.n_frames)]
    sparse_only.optimizeFrames(pb.corr, None, pb.n_frames, pb.H, pb.W, d, None, n, p2, pb.K)
    assert max(max(S.pose_error(poses_nodepth[k], p2[k])) for k in range(pb.n_frames)) < 2e-6


def test_strided_tensors_are_refused(gpu):
    from bundletrack_amd.optimizer import _dev_ptr
    t = gpu.torch.zeros((4, 6, 8, 4), device=gpu.dev)
    assert _dev_ptr(t, "zn") == t.data_ptr()
    with pytest.raises(ValueError, match="contiguous"):
        _dev_ptr(t.permute(0, 2, 1, 3), "zn")
    with pytest.raises(ValueError, match="float32"):
        _dev_ptr(t.double(), "zn")


def test_workspace_on_its_own_stream_orders_with_torch(gpu, small_problem_masked):
    """A workspace on a private non-blocking stream (btba_workspace_create) next to torch's stream: wait_stream orders the
    solve after the uploads, signal_stream orders torch's read-back after the solve -- no host synchronisation in between.
    Same poses as the default workspace (which runs on torch's stream), bit for bit."""
    from bundletrack_amd.optimizer import Workspace, build_cache_zn
    pb = small_problem_masked
    torch = gpu.torch
    d, n = upload_frames(gpu, pb)
    zn, _, _ = build_cache_zn(gpu.ws, d, n, pb.H, pb.W, pb.K, 4.0)
    gpu.ws.sync()
    outs = []
    for own in (False, True):
        ws = Workspace(use_torch_stream=False) if own else gpu.ws
        bs = gpu.BatchSolver(ws)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):                      # producer and consumer live on a third stream
            corr_d, offs_d, mx, poses_d = batch_inputs(gpu, bs, zn.cpu().numpy()[None], None, [pb.corr], [pb.poses_init])
            zn_b = zn[None].clone()
            ws.wait_stream(side.cuda_stream)
            bs.solve_zn(zn_b, pb.H, pb.W, pb.K, corr_d, offs_d, mx, poses_d)
            ws.signal_stream(side.cuda_stream)
            out = poses_d.to("cpu", non_blocking=False)
        outs.append(out.numpy()[0].copy())
        torch.cuda.synchronize()
    assert np.isfinite(outs[0]).all() and not np.array_equal(outs[0], pb.poses_init)
    assert np.array_equal(outs[0], outs[1])


def test_trusted_pair_major_upload_and_its_fallback(gpu, small_problem_masked):
    """optimizeFrames with n_match_per_pair (what Bundler::optimizeGPU passes, Bundler.cpp:298-323): the array goes to the
    device unchecked by the host and the sparse sweep verifies every entry against the pair of its segment.  Same bits as
    the host-bucketed path; a shuffled array, wrong lengths, invalid entries and out-of-range indices all end where the
    host path ends."""
    pb = small_problem_masked
    N = pb.n_frames
    d, n = upload_frames(gpu, pb)
    opt = gpu.OptimizerGpu(workspace=gpu.ws)

    def run(corr, nm):
        poses = pb.poses_init.copy()
        opt.optimizeFrames(corr, nm, N, pb.H, pb.W, d, None, n, poses, pb.K)
        return poses

    base = run(pb.corr, None)                                   # host pass over the array (no lengths given)
    assert np.array_equal(run(pb.corr, pb.n_match_per_pair), base)          # trusted: same offsets, same order, same bits
    rng = np.random.default_rng(5)
    shuffled = pb.corr[rng.permutation(len(pb.corr))]
    by_host = run(shuffled, None)
    assert np.array_equal(run(shuffled, pb.n_match_per_pair), by_host)      # lengths add up, order does not: device flag -> host bucketing
    wrong = pb.n_match_per_pair.copy(); wrong[0] += 1
    assert np.array_equal(run(pb.corr, wrong), base)                       # lengths do not add up: never trusted
    shifted = pb.n_match_per_pair.copy(); shifted[0] -= 1; shifted[1] += 1  # sum right, one boundary off by one
    assert np.array_equal(run(pb.corr, shifted), base)
    holes = pb.corr.copy()
    holes["imgIdx_i"][::7] = 0xFFFFFFFF                                    # invalid entries stay where they are on the trusted path
    r_host, r_trust = run(holes, None), run(holes, pb.n_match_per_pair)
    for k in range(N):
        rr, tt = S.pose_error(r_host[k], r_trust[k])
        assert rr < 2e-5 and tt < 2e-5                                     # same entries per pair in the same order, other lane / chunk assignment: fp32 round-off (measured 1.6e-6)
    bad = pb.corr.copy(); bad["imgIdx_j"][3] = N + 2
    for nm in (None, pb.n_match_per_pair):
        with pytest.raises(_lib.BtbaError) as e:
            run(bad, nm)
        assert e.value.status == _lib.BTBA_EINVAL


def test_24_byte_correspondences_are_bit_identical(gpu, small_problem_masked, small_problem):
    """Device-resident correspondences without their frame indices (btba_pack_correspondences24, 24 B instead of EntryJ's 32 B): the
    sparse sweep reads the same positions in the same order, so every pose is the SAME BITS as on the EntryJ array -- ragged segments,
    empty pairs and invalid entries (which keep their place) included; the packer flags an array that is not pair-major; and the keyed
    correspondence pool of btba_optimize_frames_keyed, which stores 24-byte segments and lets the sweeps read them in place, gives the
    bits of the plain call, falls back to host bucketing on a shuffled array and recovers afterwards."""
    pbs = [small_problem_masked, small_problem]
    N = pbs[0].n_frames
    bs = gpu.BatchSolver(gpu.ws)
    corr_list = []
    for b, pb in enumerate(pbs):
        c = pb.corr.copy()
        if b == 0:
            c["imgIdx_i"][::9] = 0xFFFFFFFF                                  # invalid entries
            c = c[~((c["imgIdx_i"] == 0) & (c["imgIdx_j"] == 2))]            # an empty pair -> ragged segments
        corr_list.append(c)
    corr, offs, mx = bs.pack_correspondences(corr_list, N)
    # invalid entries dropped by the host packer: put some back IN PLACE so that the device array holds holes
    corr[1]["imgIdx_i"][5:40:4] = 0xFFFFFFFF
    zn = gpu.torch.from_numpy(np.stack([S.compact_cache(pb) for pb in pbs])).to(gpu.dev)
    corr_d = gpu.torch.from_numpy(corr.view(np.uint8).reshape(len(pbs), -1, 32)).to(gpu.dev)
    offs_d = gpu.torch.from_numpy(offs.astype(np.int32)).to(gpu.dev)
    p0 = gpu.torch.from_numpy(np.stack([pb.poses_init for pb in pbs])).to(gpu.dev)
    pa, pb_ = p0.clone(), p0.clone()
    bs.solve_zn(zn, pbs[0].H, pbs[0].W, pbs[0].K, corr_d, offs_d, mx, pa)
    c24, flag = bs.pack_correspondences24(corr_d, offs_d, mx, N, check_order=True)
    bs.solve_zn(zn, pbs[0].H, pbs[0].W, pbs[0].K, None, offs_d, mx, pb_, aux={"corr24": c24}, corr_stride=corr_d.shape[1])
    gpu.ws.sync()
    assert int(flag.cpu()[0]) == 0
    assert np.array_equal(pa.cpu().numpy().view(np.uint32), pb_.cpu().numpy().view(np.uint32))
    assert not np.array_equal(pa.cpu().numpy(), p0.cpu().numpy())
    # layout: groups of 64 entries x three planes of float2; entry E = b * stride + e sits at [E // 64, :, E % 64]
    stride = corr.shape[1]
    got = c24.cpu().numpy().transpose(0, 2, 1, 3).reshape(-1, 6)[: len(pbs) * stride].reshape(len(pbs), stride, 6)
    valid = corr["imgIdx_i"] != 0xFFFFFFFF
    assert np.array_equal(got[..., :3][valid], corr["pos_i"][valid]) and np.array_equal(got[..., 3:][valid], corr["pos_j"][valid])
    filled = np.zeros(corr.shape, bool)
    for b in range(len(pbs)):
        filled[b, : int(offs[b, -1])] = True
    assert (got[..., 0].view(np.uint32)[filled & ~valid] == 0xFFFFFFFF).all()
    # a pair-major array whose entries were swapped between two segments: flagged
    sw = corr.copy(); sw[0, [0, int(offs[0, 3])]] = sw[0, [int(offs[0, 3]), 0]]
    _, flag2 = bs.pack_correspondences24(gpu.torch.from_numpy(sw.view(np.uint8).reshape(len(pbs), -1, 32)).to(gpu.dev), offs_d, mx, N, check_order=True)
    gpu.ws.sync()
    assert int(flag2.cpu()[0]) == 1

    # the keyed pool (24-byte segments read in place)
    from bundletrack_amd.optimizer import Workspace
    pb = small_problem_masked
    d, n = upload_frames(gpu, pb)
    plain = gpu.OptimizerGpu(workspace=gpu.ws)
    ws2 = Workspace()
    ws2.set_option(_lib.OPT_KEYED_CORR_MIN_BYTES, 0)
    keyed = gpu.OptimizerGpu(workspace=ws2, keyed_correspondences=True)
    keys = np.arange(100, 100 + N, dtype=np.uint64)

    def run(opt, corr_in, nm, **kw):
        poses = pb.poses_init.copy()
        opt.optimizeFrames(corr_in, nm, N, pb.H, pb.W, d, None, n, poses, pb.K, **kw)
        return poses, opt.last_stats
    base, _ = run(plain, pb.corr, pb.n_match_per_pair)
    k1, st1 = run(keyed, pb.corr, pb.n_match_per_pair, frame_keys=keys)
    k2, st2 = run(keyed, pb.corr, pb.n_match_per_pair, frame_keys=keys)
    assert np.array_equal(k1, base) and np.array_equal(k2, base)
    assert st1["corr_pairs_uploaded"] == N * (N - 1) // 2 and st2["corr_pairs_uploaded"] == 0
    holes = pb.corr.copy(); holes["imgIdx_i"][::7] = 0xFFFFFFFF
    kh, _ = run(keyed, holes, pb.n_match_per_pair, frame_keys=keys + 1000)      # other keys: fresh segments with invalid entries in place
    ph, _ = run(plain, holes, pb.n_match_per_pair)
    assert np.array_equal(kh, ph)
    shuffled = pb.corr[np.random.default_rng(5).permutation(len(pb.corr))]
    ks, sts = run(keyed, shuffled, pb.n_match_per_pair, frame_keys=keys + 2000)  # not pair-major: the packer's flag -> host bucketing, pool dropped
    ps, _ = run(plain, shuffled, None)
    assert np.array_equal(ks, ps)
    k3, st3 = run(keyed, pb.corr, pb.n_match_per_pair, frame_keys=keys)         # ... and the pool works again afterwards
    assert np.array_equal(k3, base) and st3["corr_pairs_uploaded"] == N * (N - 1) // 2


def test_atomic_reduction_mode(gpu, oracle, small_problem_masked):
    """BTBA_REDUCE_ATOMIC -- the reference's own way of summing (float atomics in arrival order, SolverBundlingDenseUtil.h:217-285): sweep
    workgroups add into one record per frame pair, k_system_solve reads and clears it.  No trace exists in this mode (BTBA_EINVAL with
    BTBA_FLAG_TRACE), so the rule is on the poses: against the exactly-summed oracle the atomic result may be off by at most
    max(1e-4, 3 x the oracle's own spread between its two summation orders) -- what the deterministic mode is held to -- and it must stay
    within that of the deterministic result; repeated runs may differ in the last bits (reported), never by more than that spread either."""
    cases = [("masked K=4", small_problem_masked, dict()), ("c3-size", S.make_problem(15, 2000, S.config_seed(5, 3), background=True, full_res=False), dict())]
    for name, pb, kw in cases:
        N = pb.n_frames
        cam, nrm, intr = oracle_cache(oracle, pb)[:3] if pb.depth is not None else S.analytic_cache(pb)
        ref = oracle.solve(cam, nrm, intr, pb.corr, pb.poses_init)                                   # exactly rounded sums
        seq = oracle.solve(cam, nrm, intr, pb.corr, pb.poses_init, params=oracle.default_params(accum_mode=0))     # sequential fp32 sums
        spread = max(max(S.pose_error(ref.poses[k], seq.poses[k])) for k in range(N))
        bound = max(1e-4, 3.0 * spread)
        zn = gpu.torch.from_numpy(S.compact_cache(pb)[None]).to(gpu.dev) if pb.depth is None else None
        outs = {}
        for mode in (_lib.REDUCE_DETERMINISTIC, _lib.REDUCE_ATOMIC, _lib.REDUCE_ATOMIC, _lib.REDUCE_ATOMIC):
            bs = gpu.BatchSolver(gpu.ws, reduction_mode=mode)
            corr, offs, mx = bs.pack_correspondences([pb.corr], N)
            corr_d = gpu.torch.from_numpy(corr.view(np.uint8).reshape(1, -1, 32)).to(gpu.dev)
            offs_d = gpu.torch.from_numpy(offs.astype(np.int32)).to(gpu.dev)
            poses_d = gpu.torch.from_numpy(pb.poses_init[None].copy()).to(gpu.dev)
            if zn is not None:
                bs.solve_zn(zn, pb.H, pb.W, pb.K, corr_d, offs_d, mx, poses_d)
            else:
                bs.solve(gpu.torch.from_numpy(cam[None]).to(gpu.dev), gpu.torch.from_numpy(nrm[None]).to(gpu.dev), intr, corr_d, offs_d, mx, poses_d)
            gpu.ws.sync()
            outs.setdefault(mode, []).append(poses_d.cpu().numpy()[0])
        det, ato = outs[_lib.REDUCE_DETERMINISTIC][0], outs[_lib.REDUCE_ATOMIC]
        for a in ato:
            assert np.isfinite(a).all()
            e_ref = max(max(S.pose_error(a[k], ref.poses[k])) for k in range(N))
            e_det = max(max(S.pose_error(a[k], det[k])) for k in range(N))
            assert e_ref < bound and e_det < bound, (name, e_ref, e_det, bound)
        run_to_run = max(max(S.pose_error(ato[0][k], a[k])) for k in range(N) for a in ato[1:])
        print(f"{name}: atomic vs oracle {max(max(S.pose_error(ato[0][k], ref.poses[k])) for k in range(N)):.2e}, vs deterministic "
              f"{max(max(S.pose_error(ato[0][k], det[k])) for k in range(N)):.2e}, run to run {run_to_run:.2e}; oracle's own summation-order spread {spread:.2e}")
        assert run_to_run < bound
    # a trace cannot be asked for in this mode
    bs = gpu.BatchSolver(gpu.ws, reduction_mode=_lib.REDUCE_ATOMIC)
    corr, offs, mx = bs.pack_correspondences([small_problem_masked.corr], small_problem_masked.n_frames)
    cam, nrm, intr = oracle_cache(oracle, small_problem_masked)[:3]
    with pytest.raises(_lib.BtbaError) as e:
        bs.solve(gpu.torch.from_numpy(cam[None]).to(gpu.dev), gpu.torch.from_numpy(nrm[None]).to(gpu.dev), intr,
                 gpu.torch.from_numpy(corr.view(np.uint8).reshape(1, -1, 32)).to(gpu.dev), gpu.torch.from_numpy(offs.astype(np.int32)).to(gpu.dev), mx,
                 gpu.torch.from_numpy(small_problem_masked.poses_init[None].copy()).to(gpu.dev), trace=True)
    assert e.value.status == _lib.BTBA_EINVAL


def test_randomised_windows_are_explained(gpu, oracle):
    """tests/tools/fuzz_parity.py's first cases through the drop-in boundary AND the traced batch entry: window sizes 2 ... 9, 0 ... 500
    correspondences per pair with emptied and thinned pairs, masked and fully valid frames.  Every result is finite, and every iterate
    that leaves the 1e-4 bar is explained the way tests/helpers.py::check_parity_with_decisions demands (a differing accept / guard decision
    before it, or the oracle's own summation-order spread): most of these windows are far weaker than a tracker's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    n_tight = 0
    for rec in fz.run_cases(14, explain_always=True):
        assert rec["finite"], rec
        assert rec["unexplained_iterates"] == [], rec
        # round 6, the licence: no iterate of the traced solve and not the boundary's own result further from the reference's forward / IEEE run than
        # max(1e-4, 3 x what the reference's own code does to itself on that window under other atomic orders and its own fast-math flags)
        if "beyond_reference_spread" in rec:
            assert rec["beyond_reference_spread"] == [] and not rec["boundary_final_beyond_reference_spread"], rec
        n_tight += max(rec["per_iterate"]) < 1e-4
    assert n_tight >= 7               # the well-posed half of the draw holds the plain 1e-4 bar on every iterate


@pytest.mark.parametrize("K", [2, 3, 6, 7, 11, 12, 15, 17, 18, 21, 22, 25, 30, 31], ids=lambda k: f"K{k}")
def test_small_solve_kernel_agrees_with_the_legacy_kernel(gpu, K):
    """k_solve_small (round 5, BTBA_OPT_SOLVE_SMALL = 1, the default for windows of <= 21 frames) against k_system_solve (0) on the same inputs: the
    same sums in another order.  Window sizes at both ends of each of its four instantiations (8 x 4 / 8 / 12 / 16 matrix columns per row group:
    <= 6 / 11 / 17 / 21 frames) and, round 6, of k_solve_mid (22 ... 31 frames: matrix rows gathered into registers, btba_solve_mid.hpp), object-masked frames, two instances per batch (one partial per sum at K >= 15, several below), per iterate:
    system matrix, right-hand side, Jacobi diagonal and accepted-pixel counts of the first linearisation to round-off, iterates within the 1e-4 bar
    while the decisions are identical."""
    from helpers import first_decision_divergence
    insts = [S.make_problem(K, 40, seed=500 + 10 * K + b, background=False, full_res=False) for b in range(2)]
    zn = np.stack([S.compact_cache(pb) for pb in insts])
    pb0 = insts[0]
    bs = gpu.BatchSolver(gpu.ws)
    corr, offs, mx = bs.pack_correspondences([pb.corr for pb in insts], K)
    zn_d = gpu.torch.from_numpy(zn).to(gpu.dev)
    corr_d = gpu.torch.from_numpy(corr.view(np.uint8).reshape(2, -1, 32)).to(gpu.dev); offs_d = gpu.torch.from_numpy(offs.astype(np.int32)).to(gpu.dev)
    views = {}
    try:
        for opt in (0, 1):
            gpu.ws.set_option(_lib.OPT_SOLVE_SMALL, opt)
            poses_d = gpu.torch.from_numpy(np.stack([pb.poses_init for pb in insts]).astype(np.float32)).to(gpu.dev)
            views[opt] = bs.trace_view(bs.solve_zn(zn_d, pb0.H, pb0.W, pb0.K, corr_d, offs_d, mx, poses_d, trace=True))
    finally:
        gpu.ws.set_option(_lib.OPT_SOLVE_SMALL, 1)
    a, b = views[0], views[1]
    for inst in range(2):
        A0, A1 = a.A[inst, 0], b.A[inst, 0]
        assert np.abs(A0 - A1).max() <= 2e-5 * np.abs(A0).max(), "system matrix of the first linearisation"
        assert np.abs(a.rhs[inst, 0] - b.rhs[inst, 0]).max() <= 1e-4 * max(np.abs(a.rhs[inst, 0]).max(), 1e-6)
        assert np.abs(a.precond[inst, 0] - b.precond[inst, 0]).max() <= 1e-5 * np.abs(a.precond[inst, 0]).max()
        assert np.array_equal(a.dense_pair[inst, 0, :, 27], b.dense_pair[inst, 0, :, 27])
        div = first_decision_divergence(b.pcg_scalars[inst], b.dense_pair[inst][..., 27], a.pcg_scalars[inst], np.rint(a.dense_pair[inst][..., 27]))
        first = div[0] if div is not None else a.T_after.shape[1]
        for it in range(first):
            for k in range(K):
                r, t = S.pose_error(a.T_after[inst, it, k], b.T_after[inst, it, k])
                assert r < 1e-4 and t < 1e-4, f"K={K} instance {inst} iterate {it} frame {k}: {r:.2e} rad / {t:.2e} m (first differing decision: {div})"
        assert np.isfinite(b.T_after[inst]).all()
