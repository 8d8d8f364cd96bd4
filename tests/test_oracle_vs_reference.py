"""The oracle against THE REFERENCE ITSELF: wenbowen123/BundleTrack's header-only device functions (SE(3) helpers,
4x4 inverse, bilinear taps, Huber, the sparse residual / Jacobian-transpose / matrix-free operator, the dense
projective association and point-to-plane rows) compiled for the CPU where they lie under /root/reference
(oracle/ref_driver.cpp, oracle/ref_shim/, `make -C oracle ref` -> oracle/_ref/libbtba_ref.so) and called on the same
inputs as oracle/btba_oracle.c.  Skipped where neither the built library nor the reference checkout exists."""
import os

import numpy as np
import pytest

from bundletrack_amd import synthetic as S
from oracle import reference as R

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libbtba_ref.so not built and /root/reference absent")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def first_iterate_transforms(oracle, poses):
    """What the solver linearises about in its first iteration: T = Exp(Log(pose)) (SBA.cpp:106,115 -> SolverBundling.cu:890-897)."""
    return np.stack([oracle.pose_to_matrix(*oracle.matrix_to_pose(P)) for P in np.asarray(poses, np.float32)])


def test_se3_helpers_bit_exact(oracle):
    rng = np.random.default_rng(0)
    cases = [np.zeros(3), [1e-5, 0, 0], [9e-5, 3e-5, 0], [1e-3, -1e-3, 2e-4], [0.03, 0.01, -0.02], [0.7, -0.4, 0.2], [2.0, 1.5, -1.0],
             [3.1, 0.2, 0.1], [0, 3.14159, 0], [1e-4, 1e-4, 1e-4]] + list(rng.uniform(-np.pi, np.pi, size=(300, 3))) + list(rng.normal(scale=1e-3, size=(100, 3)))
    for w in cases:
        t = rng.uniform(-1, 1, 3)
        Mo, Mr = oracle.pose_to_matrix(w, t), R.pose_to_matrix(w, t)
        assert np.array_equal(bits(Mo), bits(Mr)), w                         # every Exp branch: theta^2 < 1e-8, < 1e-6, general
        ro, to = oracle.matrix_to_pose(Mr); rr, tr = R.matrix_to_pose(Mr)
        assert np.array_equal(bits(ro), bits(rr)) and np.array_equal(bits(to), bits(tr)), w      # every Log branch incl. near pi
        io, ir = oracle.mat4_inverse(Mr), R.mat4_inverse(Mr)
        assert np.array_equal(io, ir)                                         # value-equal (the two differ in the sign of exact zeros)
        dW, dT = rng.normal(scale=0.01, size=3), rng.normal(scale=0.01, size=3)
        uo, ur = oracle.lie_update(dW, dT, w, t), R.lie_update(dW, dT, w, t)
        assert np.array_equal(bits(uo[0]), bits(ur[0])) and np.array_equal(bits(uo[1]), bits(ur[1]))
    for _ in range(50):                                                       # a general (non-rigid) 4x4 through the cofactor inverse
        M = rng.normal(size=(4, 4)).astype(np.float32)
        assert np.array_equal(oracle.mat4_inverse(M), R.mat4_inverse(M))


def test_jacobians_bilinear_huber_bit_exact(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        A = oracle.pose_to_matrix(rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)); D = oracle.pose_to_matrix(rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3))
        p = rng.uniform(-1, 1, 3)
        for which in ("I", "J"):
            assert np.array_equal(bits(oracle.lie_deriv(which, A, D, p)), bits(R.lie_deriv(which, A, D, p)))
    img = rng.uniform(0.2, 2.0, size=(9, 11, 4)).astype(np.float32)
    img[2, 3] = 0; img[5, 5] = 0                                              # zero (invalid) taps are blended in, ICPUtil.h:83-110
    MINF = np.float32(-np.inf)
    for x, y in [(0, 0), (10, 8), (3.3, 2.2), (2.5, 1.5), (4.7, 4.9), (-0.4, 3), (10.4, 8.3), (-1.2, 2), (3, 8.9), (11.2, 3), (9.99, 7.99), (0.0, 8.0)] + \
            [tuple(v) for v in rng.uniform(-1.5, 11.5, size=(300, 2))]:
        ok, vo = oracle.bilinear4(x, y, img)
        vr = R.bilinear4(x, y, img)
        if ok:
            assert np.array_equal(bits(vo), bits(vr)), (x, y)
        else:
            assert vr[0] == MINF, (x, y)                                      # the reference's "no tap in the image" marker
    for e in [0.0, 1e-9, 2.4e-5, 2.5e-5, 2.6e-5, 1e-4, 0.3, 7.0] + list(rng.uniform(0, 1e-3, 100)):
        assert np.float32(oracle.huber_weight(e, 0.005)) == R.huber(e, 0.005)[1]


@pytest.mark.parametrize("K,m,seed", [(4, 150, 3), (6, 300, 31), (10, 1000, 32)])
def test_sparse_term_matches_reference(oracle, K, m, seed):
    """-J^T W r, the Jacobi diagonal and the matrix-free J^T J p: the reference's evalMinusJTFDevice / applyJDevice /
    applyJTDevice against the oracle.  The reference adds several hundred terms per frame sequentially in fp32 (table
    order), the canonical oracle carries the same fp32 terms in a double: agreement to the fp32 sum's own round-off."""
    pb = S.make_problem(K, m, seed, background=True, full_res=False)
    T = first_iterate_transforms(oracle, pb.poses_init)
    tr = oracle.solve(*S.analytic_cache(pb)[:3], pb.corr, pb.poses_init, params=oracle.default_params(weight_dense_depth=0.0, n_gn_iters=1))
    rhs_r, prec_r = R.sparse_rhs(pb.corr, T)
    scale = np.abs(rhs_r).max()
    assert np.abs(tr.rhs[0].reshape(K, 6) - rhs_r).max() <= 1e-5 * scale
    po, pr = tr.precond[0].reshape(K, 6)[1:], prec_r[1:]
    assert np.abs(po - pr).max() <= 1e-5 * np.abs(pr).max()
    rng = np.random.default_rng(seed)
    p = rng.normal(size=(K, 6)).astype(np.float32); p[0] = 0
    ao, ar = oracle.sparse_apply(pb.corr, T, p), R.sparse_apply(pb.corr, T, p)
    assert np.abs(ao - ar).max() <= 1e-5 * np.abs(ar).max()
    # invalid entries are skipped by both
    corr = pb.corr.copy(); corr["imgIdx_i"][::7] = 0xFFFFFFFF
    tr2 = oracle.solve(*S.analytic_cache(pb)[:3], corr, pb.poses_init, params=oracle.default_params(weight_dense_depth=0.0, n_gn_iters=1))
    rhs_r2, _ = R.sparse_rhs(corr, T)
    assert np.abs(tr2.rhs[0].reshape(K, 6) - rhs_r2).max() <= 1e-5 * scale


@pytest.mark.parametrize("K,seed,bg", [(3, 41, True), (4, 42, False), (5, 43, True)])
def test_dense_term_matches_reference(oracle, K, seed, bg):
    """findDenseCorr + Huber + computeJacobianBlockRow_i/j + addToLocalSystemBrute run by the reference's own code over
    every pixel of every (target < source) pair, against the oracle's dense system of the first Gauss-Newton iterate:
    the same accepted pixels, the same matrix and right-hand side up to fp32 summation order."""
    pb = S.make_problem(K, 10, seed, background=bg, full_res=False)
    campos, normals, intr = S.analytic_cache(pb)
    tr = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(weight_sparse=0.0, n_gn_iters=1))
    T = first_iterate_transforms(oracle, pb.poses_init)
    Tinv = np.stack([oracle.mat4_inverse(T[k]) for k in range(K)])
    pairs = oracle.target_lower_pairs(K)
    JtJ, Jtr, cnt = R.dense_system(campos, normals, intr, T, Tinv, pairs)
    assert np.array_equal(cnt, tr.dense_count[0][:len(pairs)]) and cnt.sum() > 500
    lower = np.tril(JtJ)                                                      # FlipJtJ (SolverBundling.cu:49-59): upper <- lower
    flipped = lower + np.tril(lower, -1).T
    n = 6 * K
    Jo = tr.dense_JtJ[0].reshape(n, n)
    assert np.abs(Jo - flipped).max() <= 2e-5 * np.abs(flipped).max()
    # J^T r adds tens of thousands of signed terms sequentially in fp32 on the reference side (atomicAdd order = pixel order here)
    assert np.abs(tr.dense_Jtr[0] - Jtr).max() <= 1e-4 * np.abs(Jtr).max()


def test_ransac_hypothesis_and_vote_match_reference(oracle):
    """The reference's procrustesKernel (with its pasted approximate 3x3 SVD: 4 Jacobi sweeps, rsqrt-based Givens angles)
    and evalPoseKernel, compiled for the CPU, against oracle/btba_oracle_ransac.c (exact Kabsch).  Same conventions --
    the typical difference is 1e-6 -- and wherever the two differ more, the oracle's fit has the smaller residual: the
    reference's SVD is the inaccurate side (heavy tail on 3-point samples, whose correlation matrix has rank 2).
    For a given pose the two inlier lists are identical (same distance formula, same `dist > thres => reject` rule)."""
    from test_oracle_ransac import planted
    rng = np.random.default_rng(8)

    def fit_residual(pose, P, Q):
        return float((((P.astype(np.float64) @ pose[:3, :3].astype(np.float64).T + pose[:3, 3]) - Q) ** 2).sum())

    for n, med_bound, max_bound in ((3, 1e-5, None), (5, 1e-5, None), (40, 1e-6, 2e-3), (400, 1e-6, 5e-4)):
        diffs = []
        for _ in range(120):
            P, Q, T, _ = planted(rng, n, 0.0, noise=0.001)
            ok_r, pose_r = R.procrustes(P, Q)
            ok_o, pose_o, gap = oracle.procrustes(P, Q)
            assert ok_o
            if not ok_r:                       # "R is not valid": the reference's SVD gave up
                continue
            diffs.append(np.abs(pose_r - pose_o).max())
            ro, rr = fit_residual(pose_o, P, Q), fit_residual(pose_r, P, Q)
            assert ro <= rr * (1 + 1e-3) + 1e-9                                    # the oracle is never the worse fit ...
            if diffs[-1] > 1e-3:
                assert ro < rr                                                     # ... and where the two really differ, it is the better one
        diffs = np.array(diffs)
        print(f"procrustes n={n}: reference vs oracle pose entries: median {np.median(diffs):.1e}, 95 % {np.percentile(diffs, 95):.1e}, max {diffs.max():.1e}")
        assert np.median(diffs) < med_bound and (max_bound is None or diffs.max() < max_bound)
    for n, frac in ((60, 0.3), (500, 0.5)):
        P, Q, T, mask = planted(rng, n, frac)
        res = oracle.ransac_pair(P, Q, 200, 0.01, seed=4)
        ids_ref = R.eval_pose(P, Q, res["best_pose"], 0.01)
        assert np.array_equal(ids_ref, res["inlier_ids"]) and np.array_equal(ids_ref, np.nonzero(mask)[0])
        for t in (0, 7, 63):                   # arbitrary (bad) hypotheses too: same list, point for point
            pose = np.eye(4, dtype=np.float32); pose[:3] = res["poses"][t][:3]
            want = [i for i in range(n) if not (np.float32(np.sqrt(np.float32(((Q[i].astype(np.float32) - (pose[:3, :3] @ P[i] + pose[:3, 3]).astype(np.float32)) ** 2).sum()))) > np.float32(0.01))]
            got = R.eval_pose(P, Q, pose, 0.01)
            assert abs(len(got) - len(want)) <= 1


# ---- the reference's WHOLE solver ------------------------------------------------------------------------------
# solveBundlingStub and every kernel under it (BuildDenseSystem incl. addToLocalSystem, FlipJtJ, PCGInit, the five PCG
# kernels, the dense mat-vec, computeLieUpdate, convertLiePosesToMatricesCU, the frame->correspondence table), compiled
# for the CPU and executed thread by thread (oracle/ref_shim/cuda_runtime.h btba_emulate, oracle/ref_solver_wrap.h).

@pytest.mark.parametrize("case", [
    dict(K=4, m=150, seed=7, bg=False, wd=1.0, tol=1e-5),       # object mask, feature + dense
    dict(K=5, m=200, seed=8, bg=False, wd=1.0, tol=1e-5),
    dict(K=3, m=0, seed=9, bg=False, wd=1.0, tol=2e-3),         # dense term alone: no Jacobi diagonal (it comes from the feature term), weakly conditioned
    dict(K=6, m=300, seed=9, bg=False, wd=0.0, tol=1e-5),       # feature term alone
    dict(K=4, m=150, seed=7, bg=True, wd=1.0, tol=1e-3),        # 100 %-valid K=4: weakly conditioned
], ids=lambda c: f"K{c['K']}_m{c['m']}_{'bg' if c['bg'] else 'mask'}_wd{c['wd']:g}")
def test_reference_solver_matches_oracle_per_iterate(oracle, case):
    pb = S.make_problem(case["K"], case["m"], case["seed"], background=case["bg"], full_res=False)
    campos, normals, intr = S.analytic_cache(pb)
    tr = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(weight_dense_depth=case["wd"]))
    worst = 0.0
    for n in range(1, 8):                                        # the stub has no per-iterate output: run it for 1, 2, ... 7 iterations
        P, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, n_gn=n, weight_dense=case["wd"])
        worst = max(worst, max(max(S.pose_error(P[k], tr.T_after[n - 1][k])) for k in range(case["K"])))
        assert np.array_equal(P[0], tr.T_after[n - 1][0])        # frame 0 is never moved, by either
    print(f"{case}: reference vs oracle, worst over the 7 iterates {worst:.2e}")
    assert worst < case["tol"], worst


@pytest.mark.parametrize("name,K,m,wd,config", [("c2", 10, 1000, 0.0, 2), ("c3", 15, 2000, 1.0, 3), ("c4", 30, 4000, 1.0, 4)])
def test_reference_solver_matches_oracle_at_baseline_sizes(oracle, name, K, m, wd, config):
    """BASELINE.json configs[1], configs[2] (the headline) and configs[3] (K=30, 4k corr/pair, 60-keyframe pool pruned to 30),
    same seeds as the GPU full-size tests.  c4 takes the emulated reference ~40 s (435 pairs x 19 200 pixels x 7 iterations)."""
    seed = S.config_seed(config)
    angles = S.pruned_pool_angles(60, 30, seed) if name == "c4" else None
    pb = S.make_problem(K, m, seed, background=True, full_res=False, angles=angles)
    campos, normals, intr = S.analytic_cache(pb)
    tr = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(weight_dense_depth=wd, n_threads=4))
    P, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, weight_dense=wd)
    worst = max(max(S.pose_error(P[k], tr.poses[k])) for k in range(K))
    print(f"{name}: reference vs oracle after 7 GN x 5 PCG: {worst:.2e}")
    assert worst < 1e-4, worst


@pytest.mark.parametrize("N,m,seed", [(40, 30, 96), (85, 12, 95)])
def test_reference_solver_matches_oracle_on_large_windows(oracle, N, m, seed):
    """Windows above the 31 frames the HIP path keeps in LDS, up to the reference's MAX_NUM_IMAGES = 85 (GlobalDefines.h:8),
    on 32 x 24 caches: same problems as the GPU large-window tests."""
    Ks = S.NOCS_K.copy(); Ks[:2] *= 0.2
    pb = S.make_problem(N, m, seed=seed, background=False, H=96, W=128, K=Ks, rot_step_deg=(5.0, 6.0))
    campos, normals, intr = S.analytic_cache(pb)
    tr = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(n_threads=4))
    P, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, weight_dense=1.0)
    worst = max(max(S.pose_error(P[k], tr.poses[k])) for k in range(N))
    print(f"N={N}: reference vs oracle after 7 GN x 5 PCG: {worst:.2e}")
    assert worst < 1e-4, worst


# ---- the reference's image kernels (CUDAImageUtil.cu), emulated the same way ------------------------------------

def _kinv4(oracle, K):
    K4 = np.eye(4, dtype=np.float32); K4[:3, :3] = np.asarray(K, np.float32)
    return oracle.mat4_inverse(K4)


@pytest.mark.parametrize("rank", [None, [0, 1, 2, 3, 4], [3, 0, 4, 1, 2]])
def test_pair_orientation_follows_the_reference_address_compare(oracle, rank):
    """FindImageImageCorr_Kernel keeps (target i, source j) iff the address of frame i's d_num_valid_points is above frame j's
    (SolverBundling.cu:25-33) and FlipJtJ_Kernel erases the cross blocks written above the diagonal (:49-59).  The reference's own
    code, run with the addresses in descending / ascending / arbitrary order, against the oracle given the corresponding explicit
    (target, source) list: the oracle's flip semantics are the reference's."""
    pb = S.make_problem(5, 300, seed=41, background=False)
    N = pb.n_frames
    caches = [oracle.build_cache(pb.depth[k], pb.normals[k], pb.K) for k in range(N)]
    campos, normals, intr = np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches]), caches[0]["intr"]
    ref, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, addr_rank=rank)
    pairs = R.pairs_from_addr_rank(rank if rank is not None else list(range(N))[::-1])
    got = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, pairs=pairs, want_trace=False).poses
    worst = max(max(S.pose_error(got[k], ref[k])) for k in range(N))
    assert worst < 1e-5, worst
    if rank is not None:
        lower = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, want_trace=False).poses
        assert max(max(S.pose_error(lower[k], ref[k])) for k in range(N)) > 3e-4       # a different orientation is a different problem


def _procrustes_cases(n_cases, seed):
    rng = np.random.default_rng(seed)
    for trial in range(n_cases):
        n = 3 if trial % 4 else int(rng.integers(3, 12))
        src = (rng.normal(size=(n, 3)) * rng.choice([1e-3, 0.01, 0.1, 1.0, 30.0])).astype(np.float32)
        ang = rng.normal(size=3); th = np.linalg.norm(ang); k = ang / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        Rm = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        dst = (src @ Rm.T + rng.normal(size=3) * 0.1 + rng.normal(size=(n, 3)) * rng.choice([0, 1e-3, 1e-2])).astype(np.float32)
        if trial % 7 == 0: dst[:, 2] = -dst[:, 2]                                        # a reflection: det(V U^T) < 0
        if trial % 11 == 0: src[2] = 2 * src[1] - src[0]; dst[2] = 2 * dst[1] - dst[0]   # collinear samples
        if trial % 13 == 0: src[:] = src[0]                                              # all points equal: S = 0
        yield src, dst


def test_reference_procrustes_restatements_are_bit_exact(oracle):
    """procrustesKernel with the reference's pasted approximate 3x3 SVD (cuda_ransac.cu:48-975, 998-1103), run by the reference's own
    code (oracle/_ref/libbtba_ref_ransac.so), against the two restatements of it: the oracle's (oracle/btba_oracle_ransac.c,
    orc_procrustes_reference) and the PRODUCT's (bundletrack_amd/csrc/btba_svd3.hpp, compiled here for the host with g++ -- the same
    header k_ransac_vote includes).  Bit for bit, including the return flag, on 3-point and n-point sets, reflections, collinear and
    coincident samples, scales from 1e-3 to 30."""
    import ctypes as C, subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so, src_cpp = os.path.join(here, "cpp", "libsvd3_host.so"), os.path.join(here, "cpp", "svd3_host.cpp")
    hdr = os.path.join(os.path.dirname(here), "bundletrack_amd", "csrc", "btba_svd3.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src_cpp), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unknown-pragmas", "-o", so, src_cpp])
    f = C.CDLL(so).procrustes_reference_host
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]; f.restype = C.c_int
    n_fail = 0
    for src, dst in _procrustes_cases(4000, 3):
        ok_r, pose_r = R.procrustes(src, dst)
        ok_o, pose_o = oracle.procrustes_reference(src, dst)
        assert ok_o == ok_r and np.array_equal(pose_o.view(np.uint32), pose_r.view(np.uint32))
        s4 = np.ascontiguousarray(np.concatenate([src, np.ones((len(src), 1), np.float32)], 1), np.float32)
        d4 = np.ascontiguousarray(np.concatenate([dst, np.ones((len(dst), 1), np.float32)], 1), np.float32)
        mine = np.zeros(16, np.float32)
        ok_m = bool(f(s4.ctypes.data, d4.ctypes.data, len(src), mine.ctypes.data))
        assert ok_m == ok_r and np.array_equal(mine.reshape(4, 4).view(np.uint32), pose_r.view(np.uint32))
        n_fail += not ok_r
    print(f"'R is not valid' returned by the reference on {n_fail} of 4000 cases")


def test_frame_cache_matches_reference_kernels(oracle):
    """CUDACache::storeFrame = convertDepthFloatToCameraSpaceFloat4 + resampleFloat4 x2 + resampleFloat + countNumValidDepth
    (CUDACache.cpp:76-88), run by the reference's kernels, against orc_build_cache: bit for bit, masked and full frames,
    an odd frame size and a non-integer downscale."""
    for pb, ds in ((S.make_problem(2, 10, seed=51, background=False), 4.0), (S.make_problem(2, 10, seed=52, background=True), 4.0),
                   (S.make_problem(2, 10, seed=53, background=True, H=37, W=53, K=S.NOCS_K * np.array([[53 / 640], [37 / 480], [1.0]])), 4.0),
                   (S.make_problem(2, 10, seed=54, background=False), 3.0)):
        for k in range(2):
            depth = pb.depth[k].copy(); depth[5:9, 7:30] = 0.05; depth[20, 20] = np.nan            # below the 0.1 m validity gate, and a NaN
            cam, nrm, dd, nv = R.store_frame(depth, pb.normals[k], _kinv4(oracle, pb.K), ds)
            o = oracle.build_cache(depth, pb.normals[k], pb.K, ds)
            assert np.array_equal(bits(o["campos"]), bits(cam)) and np.array_equal(bits(o["normals"]), bits(nrm))
            assert np.array_equal(bits(o["depth"]), bits(dd)) and o["n_valid"] == nv


def test_depth_preprocessing_matches_reference_kernels(oracle):
    """Frame::processDepth (erodeDepthMapDevice, gaussFilterDepthMapDevice x2) and Frame::depthToCloudAndNormals
    (convertDepthFloatToCameraSpaceFloat4 + computeNormals_Kernel) by the reference's kernels against the oracle."""
    rng = np.random.default_rng(3)
    for shape, bg in (((96, 128), True), ((61, 67), False), ((120, 160), False)):
        Ks = S.NOCS_K.copy(); Ks[0] *= shape[1] / 640; Ks[1] *= shape[0] / 480
        pb = S.make_problem(2, 10, seed=60 + shape[0], background=bg, H=shape[0], W=shape[1], K=Ks)
        depth = (pb.depth[0] + (pb.depth[0] > 0) * rng.normal(scale=0.002, size=shape)).astype(np.float32)
        depth[rng.random(shape) < 0.02] = 0                                                      # holes
        for params in (dict(), dict(erode_radius=2, erode_diff=0.01, erode_ratio=0.5, bf_radius=1, sigma_d=1.0, sigma_r=0.02)):
            a, b = oracle.process_depth(depth, **params), R.process_depth(depth, **params)
            assert np.array_equal(bits(a), bits(b)), (shape, params)
        filt = oracle.process_depth(depth)
        no, xo = oracle.depth_to_normals(filt, pb.K)
        nr, xr = R.depth_to_normals(filt, _kinv4(oracle, pb.K))
        assert np.array_equal(bits(xo), bits(xr))
        assert np.array_equal(bits(no), bits(nr))


def test_random_windows_oracle_vs_reference(oracle):
    """The same forty seeded random windows the GPU test runs through the HIP path (tests/test_gpu_vs_reference.py), here oracle
    against the reference's own solver: the well-conditioned class within 1e-4 (it is the bar), the weakly conditioned
    class (100 %-valid frames, the dense term alone, 40 matches per pair next to it) within the 5e-3 the reference itself is
    reproducible to there."""
    rng = np.random.default_rng(2024)
    worst_strict = worst_loose = 0.0
    n_strict = 0
    for trial in range(40):
        K = int(rng.integers(2, 10))
        m = int(rng.choice([0, 40, 150, 400]))
        bg = bool(rng.integers(0, 2))
        wd = float(rng.choice([0.0, 1.0, 1.0]))
        if m == 0 and wd == 0.0:
            wd = 1.0
        pb = S.make_problem(K, m, 5000 + trial, background=bg, full_res=False, perturb_deg=float(rng.uniform(0.5, 3.0)), perturb_m=float(rng.uniform(0.001, 0.008)))
        campos, normals, intr = S.analytic_cache(pb)
        ref, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, weight_dense=wd)
        tr = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(weight_dense_depth=wd, n_threads=4))
        err = max(max(S.pose_error(tr.poses[k], ref[k])) for k in range(K))
        if (m >= 150) and (wd == 0.0 or not bg) and K >= 3:
            worst_strict = max(worst_strict, err); n_strict += 1
            assert err < 1e-4, (trial, K, m, bg, wd, err)
        else:
            worst_loose = max(worst_loose, err)
            assert err < 5e-3, (trial, K, m, bg, wd, err)
    print(f"random windows, oracle vs reference: {n_strict} well-conditioned, worst {worst_strict:.2e}; {40 - n_strict} weakly conditioned, worst {worst_loose:.2e}")
    assert n_strict >= 6


def test_the_iterate_record_of_the_emulator_equals_stopping_the_reference_early():
    """The per-iterate record rests on the emulator's launch hook: iterate n of ONE reference solve must be bit-identical to the reference stopped after
    n + 1 iterations (round 5's way of obtaining it), and the execution-order switch must leave the forward order's bits alone."""
    pb = S.make_problem(5, 300, 28, background=False, full_res=False)
    campos, normals, intr = S.analytic_cache(pb)
    _, _, ref_T = R.solve(campos, normals, intr, pb.corr, pb.poses_init, want_iterates=True)
    for n in (0, 3, 6):
        stopped, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, n_gn=n + 1)
        assert np.array_equal(stopped, ref_T[n])
    rev, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, order="reverse")
    again, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init)
    assert np.array_equal(again, ref_T[-1]) and not np.array_equal(rev, again)


def test_the_reference_against_itself_on_a_well_conditioned_and_on_a_weak_window():
    """Round 6: the launch emulator's execution-order switch (forward | reverse | seeded shuffle of every launch's (block, thread) cells: each a legal order of
    kernels that meet only through float atomics) and the `_fm` build (a model of the reference's own -use_fast_math flags, oracle/Makefile) -- the tools that
    measure how far the reference is from ITSELF (profiles/r06/reference_self_spread.json).  A shuffled run is reproducible for its seed; on a tracker-like window
    (K = 5, 300 matches per pair, object mask) all eight runs agree to well inside the 1e-4 bar; on a dense-only two-frame window of 100 %-valid frames they do not."""
    from helpers import reference_licence
    pb = S.make_problem(5, 300, 28, background=False, full_res=False)
    campos, normals, intr = S.analytic_cache(pb)
    a, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, order="shuffle:7")
    b, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, order="shuffle:7")
    c, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, order="shuffle:8")
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    fm1, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, fastmath=True, fastmath_seed=1)
    fm2, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, fastmath=True, fastmath_seed=2)
    assert not np.array_equal(fm1, fm2)                                   # another seed = another (equally legal) libdevice
    # the fast-math library flushes denormals only INSIDE its own calls: linked with gcc's math flags it would pull in crtfastmath.o, whose constructor
    # sets flush-to-zero for the whole process when the library is loaded (numpy, the oracle and the IEEE reference library included) -- oracle/Makefile links it without them
    tiny = np.float32(1e-40)
    assert tiny * np.float32(1.0) != 0 and np.finfo(np.float32).smallest_subnormal > 0
    cum, runs = reference_licence(R, S.pose_error, campos, normals, intr, pb.corr, pb.poses_init)
    assert runs.shape[:2] == (8, 7) and cum[-1] < 1e-4 and cum[0] > 0.0, cum
    weak = S.make_problem(2, 0, 5003, background=True, full_res=False, perturb_deg=2.0, perturb_m=0.005)
    cw, nw, iw = S.analytic_cache(weak)
    cum_w, _ = reference_licence(R, S.pose_error, cw, nw, iw, weak.corr, weak.poses_init)
    print(f"reference vs reference, last iterate: tracker-like window {cum[-1]:.2e}, dense-only two-frame window {cum_w[-1]:.2e}")
    assert cum_w[-1] > cum[-1]
