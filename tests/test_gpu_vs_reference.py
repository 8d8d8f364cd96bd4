"""The HIP path against THE REFERENCE'S OWN SOLVER: wenbowen123/BundleTrack's solveBundlingStub and all its kernels,
compiled for the CPU in the build container (oracle/_ref/libbtba_ref_solver.so, see tests/test_oracle_vs_reference.py) and
shipped to the GPU box as a prebuilt file -- nothing under /root/reference is read here.  Skipped if the file is absent."""
import os

import numpy as np
import pytest

from bundletrack_amd import synthetic as S
from oracle import reference as R

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(R.SO_SOLVER), reason="oracle/_ref/libbtba_ref_solver.so not built")]


def hip_solve(pb, wd, want_trace=False):
    import torch
    from bundletrack_amd.optimizer import BatchSolver, Workspace
    dev = torch.device("cuda:0")
    bs = BatchSolver(Workspace(), weight_dense_depth=wd)
    corr, offs, mx = bs.pack_correspondences([pb.corr], pb.n_frames)
    zn = S.compact_cache(pb)   # astype() alone keeps a strided layout
    zn_d = torch.from_numpy(zn[None]).to(dev)
    corr_d = torch.from_numpy(corr.view(np.uint8).reshape(1, -1, 32)).to(dev)
    offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
    poses_d = torch.from_numpy(pb.poses_init[None].copy()).to(dev)
    tr = bs.solve_zn(zn_d, pb.H, pb.W, pb.K, corr_d, offs_d, mx, poses_d, trace=want_trace)
    bs.ws.sync()
    if want_trace:
        return poses_d.cpu().numpy()[0], bs.trace_view(tr)
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


def test_c4_matches_the_reference_solver():
    """BASELINE configs[3] at full size (K = 30, 4 000 correspondences per pair, 60-keyframe pool pruned to 30) directly against
    the reference's own solver (about 40 s of thread-by-thread emulation), not only through the oracle."""
    pb = S.make_problem(30, 4000, S.config_seed(4), background=True, full_res=False, angles=S.pruned_pool_angles(60, 30, S.config_seed(4)))
    campos, normals, intr = S.analytic_cache(pb)
    ref, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, weight_dense=1.0)
    got = hip_solve(pb, 1.0)
    worst = max(max(S.pose_error(got[k], ref[k])) for k in range(30))
    print(f"c4: HIP vs the reference's own solver: worst pose difference {worst:.2e}")
    assert worst < 1e-4, worst


@pytest.mark.parametrize("policy", ["TARGET_HIGHER", "TARGET_MORE_VALID", "EXPLICIT"])
def test_pair_orientation_policies_against_the_reference_address_compare(policy):
    """The reference orients a dense pair by comparing the ADDRESSES of the frames' d_num_valid_points allocations
    (FindImageImageCorr_Kernel, SolverBundling.cu:17-47) and FlipJtJ_Kernel (:49-59) then erases the cross block of every pair
    whose target index is above its source index.  oracle/_ref runs exactly that code with the addresses laid out in a chosen
    order; the boundary's pair policies must reproduce it: ascending addresses = TARGET_HIGHER, addresses ordered by valid-pixel
    count = TARGET_MORE_VALID, an arbitrary order = an EXPLICIT list."""
    import torch
    from bundletrack_amd import _lib
    from bundletrack_amd.optimizer import OptimizerGpu, Workspace
    dev = torch.device("cuda:0")
    # frames masked to the object with DIFFERENT valid-pixel counts (orbit + distinct semi-axes), so MORE_VALID is a real permutation
    pb = S.make_problem(5, 300, seed=41, background=False)
    N = pb.n_frames

    def thin(k, frac):                         # drop the upper part of the object in frame k: on the orbit the counts fall monotonically with k otherwise
        rows = np.nonzero((pb.depth[k] > 0).any(1))[0]
        cut = rows[0] + int(frac * len(rows))
        pb.depth[k][:cut] = 0; pb.normals[k][:cut] = 0
    thin(0, 0.5); thin(2, 0.25)
    from oracle import oracle as O            # only for the reference-ordered cache (checked bit-exact elsewhere)
    caches = [O.build_cache(pb.depth[k], pb.normals[k], pb.K) for k in range(N)]
    campos, normals, intr = np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches]), caches[0]["intr"]
    nvalid = np.array([int((c["campos"][..., 3] == 1).sum()) for c in caches])
    if policy == "TARGET_HIGHER":
        rank = np.arange(N)
        kw = dict(pair_policy=_lib.PAIRS_TARGET_HIGHER); pairs = None
    elif policy == "TARGET_MORE_VALID":
        assert len(set(nvalid.tolist())) > 2, nvalid
        order = sorted(range(N), key=lambda k: (nvalid[k], -k))        # ascending count; ties: the lower index counts as "more valid" (i < j kept)
        rank = np.empty(N, np.int64); rank[order] = np.arange(N)
        kw = dict(pair_policy=_lib.PAIRS_TARGET_MORE_VALID); pairs = None
    else:
        rank = np.array([3, 0, 4, 1, 2])
        kw = dict(pair_policy=_lib.PAIRS_EXPLICIT); pairs = R.pairs_from_addr_rank(rank)
    ref, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, addr_rank=rank)
    ref_lower, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init)
    assert max(max(S.pose_error(ref[k], ref_lower[k])) for k in range(N)) > 3e-4      # the orientation matters on this window
    opt = OptimizerGpu(workspace=Workspace(), **kw)
    d = [torch.from_numpy(pb.depth[k]).to(dev) for k in range(N)]
    n = [torch.from_numpy(pb.normals[k]).to(dev) for k in range(N)]
    poses = pb.poses_init.copy()
    opt.optimizeFrames(pb.corr, pb.n_match_per_pair, N, pb.H, pb.W, d, None, n, poses, pb.K, dense_pairs=pairs)
    worst = max(max(S.pose_error(poses[k], ref[k])) for k in range(N))
    print(f"{policy}: HIP vs the reference run with address order {rank.tolist()}: worst pose difference {worst:.2e}")
    assert worst < 1e-4, worst


def test_large_window_matches_the_reference_solver():
    """40 frames (above BTBA_MAX_FRAMES_LDS = 31: matrix in the global scratch, 16-wave PCG) on 32 x 24 caches against the
    reference's own solver, whose limit is MAX_NUM_IMAGES = 85."""
    Ks = S.NOCS_K.copy(); Ks[:2] *= 0.2
    pb = S.make_problem(40, 30, seed=96, background=False, H=96, W=128, K=Ks, rot_step_deg=(5.0, 6.0))
    campos, normals, intr = S.analytic_cache(pb)
    ref, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, weight_dense=1.0)
    got = hip_solve(pb, 1.0)
    worst = max(max(S.pose_error(got[k], ref[k])) for k in range(pb.n_frames))
    print(f"N=40: HIP vs the reference's own solver: worst pose difference {worst:.2e}")
    assert worst < 1e-4, worst


@pytest.mark.skipif(not os.path.exists(R.SO_IMAGE), reason="oracle/_ref/libbtba_ref_image.so not built")
def test_hip_frame_cache_and_preprocessing_match_the_reference_kernels():
    """btba_build_cache / btba_process_depth / btba_depth_to_normals against the reference's own CUDAImageUtil kernels
    (run on the CPU): the frame cache bit for bit, the filtered depth and the normals to fp32 round-off."""
    import torch
    from bundletrack_amd.optimizer import Workspace, build_cache, process_depth, depth_to_normals
    dev = torch.device("cuda:0")
    ws = Workspace()
    pb = S.make_problem(3, 10, seed=77, background=False)
    K4 = np.eye(4, dtype=np.float32); K4[:3, :3] = pb.K
    from oracle import oracle as O            # only for the reference-ordered 4x4 inverse of K
    Kinv = O.mat4_inverse(K4)
    d = [torch.from_numpy(pb.depth[k]).to(dev) for k in range(3)]
    n = [torch.from_numpy(pb.normals[k]).to(dev) for k in range(3)]
    campos, nrm, nvalid, intr = build_cache(ws, d, n, pb.H, pb.W, pb.K)
    ws.sync()
    for k in range(3):
        cam_r, nrm_r, _, nv_r = R.store_frame(pb.depth[k], pb.normals[k], Kinv)
        assert np.array_equal(campos[k].cpu().numpy().view(np.uint32), cam_r.view(np.uint32))
        assert np.array_equal(nrm[k].cpu().numpy().view(np.uint32), nrm_r.view(np.uint32))
        assert int(nvalid[k]) == nv_r
    rng = np.random.default_rng(1)
    Ks = S.NOCS_K.copy(); Ks[0] *= 160 / 640; Ks[1] *= 120 / 480
    small = S.make_problem(2, 10, seed=78, background=True, H=120, W=160, K=Ks)
    depth = (small.depth[0] + rng.normal(scale=0.002, size=small.depth[0].shape)).astype(np.float32)
    filt = process_depth(ws, torch.from_numpy(depth).to(dev))
    ws.sync()
    ref_filt = R.process_depth(depth)
    fg = filt.cpu().numpy()
    # the filter weights go through expf: the device's and glibc's differ in the last ulp, so the filtered depth agrees to
    # round-off (like tests/test_depth_processing.py against the oracle), with the same validity pattern
    mism = (fg == 0) != (ref_filt == 0)                                  # a pixel within an ulp of the 0.01 m mean gate may flip
    diff = np.abs(fg - ref_filt)[~mism]
    # ... and so may a neighbour's membership in a pixel's window (|c - mean| < 0.01): a handful of outputs move by a tap's weight
    assert mism.sum() <= 4 and (diff > 2e-6).sum() <= 4 and diff.max() < 2e-3, (int(mism.sum()), int((diff > 2e-6).sum()), float(diff.max()))
    K4s = np.eye(4, dtype=np.float32); K4s[:3, :3] = small.K
    nr, _ = R.depth_to_normals(fg, O.mat4_inverse(K4s))                 # same input on both sides
    ng = depth_to_normals(ws, filt, small.K)
    ws.sync()
    ng = ng.cpu().numpy()
    assert np.array_equal(ng == 0, nr == 0) and np.abs(ng - nr).max() <= 2e-6


def test_random_windows_match_the_reference_solver(oracle):
    """Forty seeded windows of random shape (2-9 frames, 0-400 matches per pair, masked or 100 %-valid frames, feature and
    dense weights on or off, perturbations up to 3 deg / 8 mm) through the HIP path and through the reference's own solver, EVERY
    Gauss-Newton iterate (the reference's iterates recorded by the launch emulator's hook, oracle/ref_solver_wrap.h).
    Well-conditioned windows (>= 150 matches per pair on an object mask, or features alone) are held to the 1e-4 bar.
    The others -- 100 %-valid frames, the dense term alone, or only 40 matches per pair next to it -- are windows the reference does
    not determine to 1e-4 ITSELF, and round 6 measures that instead of asserting it: a window on which the HIP path leaves the bar is
    run through the reference's own code under three other legal execution orders of its float atomics (SolverBundlingDenseUtil.h:217-285,
    SolverBundling.cu:575-818) and under a model of its own build flags (-use_fast_math, CMakeLists.txt:7) -- helpers.reference_licence,
    profiles/r06/reference_self_spread.json -- and the HIP iterate must lie within max(1e-4, 3 x the reference's spread so far).
    A window above 1e-4 that the reference DOES hold against itself is listed as `beyond`; it must then at least be explained by the older
    rule (a differing accept / epsilon-guard decision before the excursion, helpers.first_decision_divergence), and there may be at most two."""
    from helpers import beyond_reference_spread, check_parity_with_decisions, first_decision_divergence, reference_licence
    rng = np.random.default_rng(2024)
    worst_strict = worst_loose = 0.0
    n_strict = n_held = 0
    above, licensed, beyond = [], [], []
    n_windows = 40
    for trial in range(n_windows):
        K = int(rng.integers(2, 10))
        m = int(rng.choice([0, 40, 150, 400]))
        bg = bool(rng.integers(0, 2))
        wd = float(rng.choice([0.0, 1.0, 1.0]))
        if m == 0 and wd == 0.0:
            wd = 1.0
        pb = S.make_problem(K, m, 5000 + trial, background=bg, full_res=False, perturb_deg=float(rng.uniform(0.5, 3.0)), perturb_m=float(rng.uniform(0.001, 0.008)))
        campos, normals, intr = S.analytic_cache(pb)
        ref, _, ref_T = R.solve(campos, normals, intr, pb.corr, pb.poses_init, weight_dense=wd, want_iterates=True)
        got, tv = hip_solve(pb, wd, want_trace=True)
        assert np.isfinite(got).all()
        err = max(max(S.pose_error(got[k], ref[k])) for k in range(K))
        per_it = [max(max(S.pose_error(tv.T_after[0, it, k], ref_T[it, k])) for k in range(K)) for it in range(ref_T.shape[0])]
        strict = (m >= 150) and (wd == 0.0 or not bg) and K >= 3
        if strict:
            worst_strict = max(worst_strict, err); n_strict += 1
            assert max(per_it) < 1e-4, (trial, K, m, bg, wd, per_it)
        else:
            worst_loose = max(worst_loose, err)
            assert err < 5e-3, (trial, K, m, bg, wd, err)
        if max(per_it) < 1e-4:
            n_held += 1
            continue
        above.append((trial, K, m, 'full' if bg else 'mask', wd, float(f'{max(per_it):.1e}')))
        cum, _ = reference_licence(R, S.pose_error, campos, normals, intr, pb.corr, pb.poses_init, weight_dense=wd)
        bad, _ = beyond_reference_spread(tv.T_after[0], ref_T, cum, S.pose_error)
        if not bad:
            licensed.append((trial, float(f"{cum[-1]:.1e}")))
            continue
        # the reference holds this window tighter than the HIP path does: the older explanation must apply, and the window is counted
        beyond.append((trial, K, m, 'full' if bg else 'mask', wd, [float(f"{x:.1e}") for x in per_it], [float(f"{x:.1e}") for x in cum]))
        ora = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(weight_dense_depth=wd))
        div = first_decision_divergence(tv.pcg_scalars[0], tv.dense_pair[0][..., 27] if wd > 0 else None, ora.pcg_scalars, ora.dense_count if wd > 0 else None)
        check_parity_with_decisions(tv.T_after[0], ora.T_after, div, S.pose_error, 1e-4, 5e-3, f"window {trial} (K={K}, m={m}, bg={bg}, wd={wd})", ref_spread=cum)
    print(f"random windows: {n_strict} well-conditioned, worst {worst_strict:.2e}; {n_windows - n_strict} weakly conditioned, worst {worst_loose:.2e}; "
          f"{n_held} of {n_windows} hold 1e-4 against the reference's own solver on EVERY iterate")
    print(f"{len(above)} leave the bar (trial, K, m, frames, w_dense, worst iterate): {above}")
    print(f"of those, {len(licensed)} within 3x the reference's OWN spread (trial, the reference's spread at the last iterate): {licensed}")
    print(f"beyond the reference's own spread: {len(beyond)}: {beyond}")
    assert len(above) <= n_windows // 4                  # even in the weak class most windows agree to 1e-4
    assert len(beyond) <= 2, beyond
    assert n_strict >= 6


@pytest.mark.parametrize("name,ws,wd", [
    ("dense term switched on at the third iteration", None, [0, 0, 1, 1, 1, 1, 1]),
    ("both terms ramped", [1, 1, 0.5, 0.5, 0.25, 0.25, 1], [0.25, 0.5, 1, 1, 2, 0, 1]),
    ("sparse weight 0 in two iterations (preconditioner still from the correspondences)", [1, 0, 1, 0, 1, 1, 1], None),
])
def test_per_iteration_weights_match_the_reference_seam(name, ws, wd):
    """solveBundlingStub reads its weights PER ITERATION (parameters.weightSparse = input.weightsSparse[nIter], weightDenseDepth likewise,
    useDense = weightDenseDepth > 0: SolverBundling.cu:948-953; SBA.cpp:27-32 fills the arrays).  btba_params.weights_*_per_iter is that seam:
    the reference's own solver is run with the same arrays."""
    import torch
    from bundletrack_amd.optimizer import BatchSolver, Workspace
    pb = S.make_problem(6, 400, 77, background=False, full_res=False)
    campos, normals, intr = S.analytic_cache(pb)
    ref, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init, weights_sparse=ws, weights_dense=wd)
    const, _ = R.solve(campos, normals, intr, pb.corr, pb.poses_init)
    dev = torch.device("cuda:0")
    bs = BatchSolver(Workspace())
    bs.set_iteration_weights(sparse=ws, dense=wd)
    corr, offs, mx = bs.pack_correspondences([pb.corr], pb.n_frames)
    zn_d = torch.from_numpy(S.compact_cache(pb)[None]).to(dev)
    corr_d = torch.from_numpy(corr.view(np.uint8).reshape(1, -1, 32)).to(dev)
    offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
    poses_d = torch.from_numpy(pb.poses_init[None].copy()).to(dev)
    bs.solve_zn(zn_d, pb.H, pb.W, pb.K, corr_d, offs_d, mx, poses_d)
    bs.ws.sync()
    got = poses_d.cpu().numpy()[0]
    worst = max(max(S.pose_error(got[k], ref[k])) for k in range(pb.n_frames))
    moved = max(max(S.pose_error(ref[k], const[k])) for k in range(pb.n_frames))
    print(f"{name}: HIP vs the reference's solver {worst:.2e}; the schedule moves the result by {moved:.2e} against constant weights")
    assert worst < 1e-4, worst
    assert moved > 2e-4, "the weight schedule must matter for this to test anything"


@pytest.mark.parametrize("name,K,m,wd,seed,bg", [
    ("c3 benched instance 0", 15, 2000, 1.0, S.config_seed(5, 0), True),
    ("c3-masked", 15, 2000, 1.0, S.config_seed(3), False),        # the tracker's operating point
    ("c2", 10, 1000, 0.0, S.config_seed(2), True),
    ("c4", 30, 4000, 1.0, S.config_seed(4), True),                # K = 30: k_solve_mid; ~40 s of thread-by-thread emulation
])
def test_every_gauss_newton_iterate_against_the_reference(name, K, m, wd, seed, bg):
    """north_star's tolerance is PER Gauss-Newton iterate against the reference.  The HIP path's traced iterates T_after[n] against solveBundlingStub
    (SolverBundling.cu:931-1003, the reference's own code through the CPU launch emulator), whose unknowns after every iteration are recorded by the emulator's
    launch hook at the top of the next one (oracle/ref_solver_wrap.h: ref_solve4) -- one emulated solve per configuration (round 5: seven solves stopped after
    n = 1 ... 7 iterations, c3 only).  BASELINE configs[1], [2] (100 %-valid and object-masked) and [3]."""
    angles = S.pruned_pool_angles(60, 30, seed) if K == 30 else None
    pb = S.make_problem(K, m, seed, background=bg, full_res=False, angles=angles)
    campos, normals, intr = S.analytic_cache(pb)
    _, tv = hip_solve(pb, wd, want_trace=True)
    ref, _, ref_T = R.solve(campos, normals, intr, pb.corr, pb.poses_init, weight_dense=wd, want_iterates=True)
    assert np.array_equal(ref_T[-1], ref)
    worst = [max(max(S.pose_error(tv.T_after[0, n, k], ref_T[n, k])) for k in range(K)) for n in range(7)]
    print(f"{name}, HIP vs the reference's own solver per Gauss-Newton iterate: " + " ".join(f"{w:.1e}" for w in worst))
    assert max(worst) < 1e-4, worst
