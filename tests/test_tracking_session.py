"""a sliding K=5 window over a synthetic orbit, ~300 correspondences per pair,
driven frame by frame through bundler.Bundler (processNewFrame -> selectKeyFramesForBA -> optimizeGPU ->
checkAndAddKeyframe, src/Bundler.cpp:52-359).  CPU: the oracle is the optimiser.  GPU: the HIP path is, with
the oracle run beside it on identical inputs at every call."""
import os

import numpy as np
import pytest

from bundletrack_amd import _lib, synthetic as S
from bundletrack_amd.bundler import Bundler, FrameRef, format_pose_txt, load_pose_txt, solve_rigid_transform_between_points

from helpers import OracleOptimizer, ParityOptimizer


def run_session(optimizer, n_frames, tmp_path=None, to_device=None, max_BA_frames=5, persistent_frame_cache=False, ransac=None):
    seq = S.SyntheticSequence(n_frames=n_frames, seed=S.config_seed(1))
    fm = S.SyntheticFeatureManager(seq, corr_per_pair=300, ransac=ransac)
    bundler = Bundler(optimizer, fm, seq.K, seq.H, seq.W, window_size=5, max_BA_frames=max_BA_frames, pose_dir=tmp_path,
                      persistent_frame_cache=persistent_frame_cache)
    errs, frames = [], []
    for k in range(n_frames):
        depth, normals = seq.render(k)
        if to_device is not None:
            depth, normals = to_device(depth), to_device(normals)
        fr = FrameRef(id=0, pose_in_model=seq.poses_gt[0].astype(np.float32), n_keypts=300, depth_gpu=depth, normal_gpu=normals)
        fm.register(fr, k)
        bundler.process_new_frame(fr)
        frames.append(fr)
        errs.append(S.pose_error(fr.pose_in_model, seq.poses_gt[k]))
        assert len(bundler.local_frames) <= max_BA_frames
        assert len(bundler.frames) <= bundler.window_size + 3
    return seq, bundler, frames, np.array(errs)


def check_session(seq, bundler, frames, errs, n_frames):
    assert bundler.n_ba_calls == n_frames - 1                       # every frame after the first is bundle-adjusted
    assert [f.id for f in frames] == list(range(n_frames))
    # keyframes: frame 0 plus frames at least min_rot away from every earlier keyframe (Bundler.cpp:185-219)
    kf = bundler.keyframes
    assert kf[0].id == 0 and len(kf) >= n_frames // 3
    for a in range(len(kf)):
        for b in range(a):
            ang = np.rad2deg(S.rotation_angle(kf[a].pose_in_model[:3, :3], kf[b].pose_in_model[:3, :3]))
            assert ang >= 10.0 - 0.6       # poses moved a little in later BA calls after the frame became a keyframe
    # tracking stays on the ground truth (1 mm feature noise, 5 % outliers, Huber): no drift over the orbit
    assert errs[:, 0].max() < np.deg2rad(0.5) and errs[:, 1].max() < 0.005, errs.max(0)


def test_c1_sliding_window_oracle(oracle, tmp_path):
    n = 24
    seq, bundler, frames, errs = run_session(OracleOptimizer(oracle), n, tmp_path=str(tmp_path))
    check_session(seq, bundler, frames, errs, n)
    # pose files: poses/<id>.txt holds ob_in_cam with 10 significant digits (Bundler.cpp:362-377)
    for k in (1, n - 1):
        M = load_pose_txt(os.path.join(str(tmp_path), "%04d.txt" % k))
        # the file is written right after the frame's own BA call; later calls may still move a keyframe's pose
        assert M.shape == (4, 4) and np.allclose(M[3], [0, 0, 0, 1])
        r, t = S.pose_error(np.linalg.inv(M), seq.poses_gt[k])
        assert r < np.deg2rad(0.5) and t < 0.003


def test_pose_txt_format_and_kabsch():
    M = np.eye(4, dtype=np.float32)
    M[0, 3], M[1, 2] = 0.123456789, -1e-5
    txt = format_pose_txt(M)
    rows = txt.rstrip("\n").split("\n")
    assert len(rows) == 4 and len({len(r) for r in rows}) == 1            # Eigen pads every column to one width
    assert rows[0].split() == ["1", "0", "0", "0.123456791"]            # float -> %.10g
    assert rows[1].split()[2] == "-9.999999747e-06"
    assert np.allclose(np.array([[float(c) for c in r.split()] for r in rows]), M, rtol=1e-9, atol=0)
    # Kabsch (Utils.cpp:180-214): recovers a rigid motion, repairs a reflection, identity on degenerate input
    rng = np.random.default_rng(0)
    P = rng.normal(size=(50, 3)).astype(np.float32)
    T = S.se3_exp(np.array([0.2, -0.1, 0.3]), np.array([0.01, 0.02, -0.03]))
    Q = P @ T[:3, :3].T + T[:3, 3]
    est = solve_rigid_transform_between_points(P, Q)
    assert np.abs(est - T).max() < 1e-5
    flat = P.copy(); flat[:, 2] = 0                                        # planar points: SVD may return a reflection
    est = solve_rigid_transform_between_points(flat, flat @ T[:3, :3].T + T[:3, 3])
    assert abs(np.linalg.det(est[:3, :3]) - 1) < 1e-4 and np.abs(est - T).max() < 1e-4
    assert np.array_equal(solve_rigid_transform_between_points(P[:2], Q[:2]), np.eye(4, dtype=np.float32))
    bad = P.copy(); bad[0, 0] = np.nan
    assert np.array_equal(solve_rigid_transform_between_points(bad, Q), np.eye(4, dtype=np.float32))


@pytest.mark.gpu
def test_c1_sliding_window_hip_vs_oracle(oracle, tmp_path):
    import torch
    from bundletrack_amd.optimizer import OptimizerGpu, Workspace
    dev = torch.device("cuda:0")
    n = 60
    from bundletrack_amd import _lib
    from bundletrack_amd.optimizer import BatchSolver, build_cache_zn
    from helpers import first_decision_divergence
    ws = Workspace()

    def classify(corr, N, H, W, depths, normals, poses_in, K):
        """The same call again with traces on both sides (compact cache + batched solve = what the boundary runs inside):
        which accept / epsilon-guard decision differs first."""
        caches = [oracle.build_cache(depths[k].cpu().numpy().reshape(H, W), normals[k].cpu().numpy().reshape(H, W, 4), K) for k in range(N)]
        ora = oracle.solve(np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches]), caches[0]["intr"], corr, poses_in)
        zn, nvalid, _ = build_cache_zn(ws, depths, normals, H, W, K, 4.0)
        bs = BatchSolver(ws)
        if int(nvalid.sum()) * 10 < N * zn.shape[1] * zn.shape[2] * 6:
            bs.params.flags |= _lib.FLAG_COMPACTION                     # btba_optimize_frames' own rule for masked frames
        c, o, mx = bs.pack_correspondences([corr], N)
        tv = bs.trace_view(bs.solve_zn(zn[None], H, W, K, torch.from_numpy(c.view(np.uint8).reshape(1, -1, 32)).to(dev), torch.from_numpy(o.astype(np.int32)).to(dev), mx,
                                       torch.from_numpy(np.asarray(poses_in, np.float32)[None].copy()).to(dev), trace=True))
        # round 6, the licence first: is the HIP result within 3x what the REFERENCE'S OWN code does to itself on this call under other legal orders of its float
        # atomics and under its own fast-math flags?  (helpers.reference_licence; oracle/_ref prebuilt -- skipped where it is absent)
        from oracle import reference as R
        if os.path.exists(R.SO_SOLVER) and os.path.exists(R.SO_SOLVER_FM):
            from helpers import reference_licence
            cam, nrm = np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches])
            cum, runs = reference_licence(R, S.pose_error, cam, nrm, caches[0]["intr"], corr, poses_in)
            e_ref = max(max(S.pose_error(tv.T_after[0, -1, k], runs[0, -1, k])) for k in range(N))
            if e_ref < max(1e-4, 3.0 * cum[-1]):
                return ("reference-spread", {"hip_vs_reference": e_ref, "reference_vs_itself": float(cum[-1])})
        div = first_decision_divergence(tv.pcg_scalars[0], tv.dense_pair[0][..., 27], ora.pcg_scalars, ora.dense_count)
        if div is None:          # no decision differs: is the oracle's own summation-order spread on this call of that size? (round-off on an ill-conditioned window)
            cam, nrm = np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches])
            seq = oracle.solve(cam, nrm, caches[0]["intr"], corr, poses_in, params=oracle.default_params(accum_mode=0))
            spread = max(max(S.pose_error(seq.poses[k], ora.poses[k])) for k in range(N))
            err = max(max(S.pose_error(tv.T_after[0, -1, k], ora.poses[k])) for k in range(N))
            return ("round-off", {"oracle_summation_spread": spread, "hip_vs_oracle": err}) if err < 3 * spread else None
        return div

    par = ParityOptimizer(OptimizerGpu(workspace=ws), OracleOptimizer(oracle), S.pose_error, classify=classify, classify_above=5e-5)
    seq, bundler, frames, errs = run_session(par, n, tmp_path=str(tmp_path), to_device=lambda a: torch.from_numpy(a).to(dev))
    assert len(par.diffs) == n - 1
    d = np.array(par.diffs)
    # every call above HALF the bar (5e-5) is classified, printed and must be explained by a decision the two sides took differently (or by
    # the oracle's own summation-order spread): where exactly the largest call lands moves from box to box with the last bits
    # (1.05e-4 ... 1.37e-4 over round 2's boxes), so the record of this box goes to gpurun_out/ for profiles/
    for call in np.nonzero(d >= 5e-5)[0]:
        print(f"BA call {call}: diff {d[call]:.2e}, first differing decision {par.divergences.get(int(call))}")
        assert par.divergences.get(int(call)) is not None, f"BA call {call} differs by {d[call]:.2e} with identical accept / guard decisions"
    try:
        if not os.environ.get("BTBA_SESSION_RECORD"):      # the per-box record is written only on request (BTBA_SESSION_RECORD=<file>): a test run leaves no files behind
            raise OSError
        import json, socket
        rec = {"host": socket.gethostname(), "calls": int(len(d)), "median": float(np.median(d)), "max": float(d.max()),
               "above_5e-5": [{"call": int(c), "diff": float(d[c]), "explained_by": str(par.divergences.get(int(c)))[:300]} for c in np.nonzero(d >= 5e-5)[0]]}
        out_file = os.environ["BTBA_SESSION_RECORD"]
        os.makedirs(os.path.dirname(os.path.abspath(out_file)), exist_ok=True)
        with open(out_file, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    # A converged window sits on the reference's PCG guard (r.z <= 1e-6 => no step, SolverBundling.cu:728-818): r.z
    # hovers at 0.9-1.1e-6 and last-bit rounding decides whether one more ~1e-4 step is taken in an iteration.  The
    # oracle's own two summation orders disagree on those calls (tests/tools/dbg_session.py); one or two flipped steps move a
    # pose by 1-2.5e-4, so those calls are held to 5e-4.
    print("BA calls: %d, median diff %.2e, over 1e-4: %s" % (len(d), np.median(d), np.round(d[d >= 1e-4], 6).tolist()))
    # (the share of calls inside 1e-4: the reference's OWN solver, run under other atomic orders and its fast-math flags, keeps 52 of these 59 calls = 0.88 within 1e-4 of
    # itself -- profiles/r06/reference_self_spread.json, set `session` -- so 0.85 is asked of the HIP path against the oracle; every call above 5e-5 was classified above)
    assert np.median(d) < 1e-5 and (d < 1e-4).mean() >= 0.85 and d.max() < 5e-4, d
    check_session(seq, bundler, frames, errs, n)


@pytest.mark.gpu
def test_session_with_persistent_frame_cache_is_identical():
    """The tracker hands Frame ids to the optimiser as cache keys: one frame is cached per BA call instead of up to
    five, and every pose of the session is bit-identical to the stateless path."""
    import torch
    from bundletrack_amd.optimizer import OptimizerGpu, Workspace
    dev = torch.device("cuda:0")
    up = lambda a: torch.from_numpy(a).to(dev)
    n = 30
    a_opt, b_opt = OptimizerGpu(workspace=Workspace()), OptimizerGpu(workspace=Workspace())
    built = []
    orig = b_opt.optimizeFrames
    def counting(*a, **k):
        r = orig(*a, **k); built.append(b_opt.last_stats["cache_frames_built"]); return r
    b_opt.optimizeFrames = counting
    _, _, fa, _ = run_session(a_opt, n, to_device=up)
    _, _, fb, _ = run_session(b_opt, n, to_device=up, persistent_frame_cache=True)
    for x, y in zip(fa, fb):
        assert np.array_equal(x.pose_in_model, y.pose_in_model)
    assert built[0] == 2 and all(c == 1 for c in built[1:]), built      # first call: frames 0 and 1; afterwards only the new frame
    # ... and with the pairs' correspondence segments kept on the device too (BTBA_FLAG_KEYED_CORR): still the same bits, and after
    # the first call only the NEW frame's pairs (window size - 1 segments) cross PCIe (the library skips the bookkeeping below 1 MB
    # of correspondences; the environment variable lowers that threshold for this small session)
    c_opt = OptimizerGpu(workspace=Workspace(), keyed_correspondences=True)
    c_opt.workspace.set_option(_lib.OPT_KEYED_CORR_MIN_BYTES, 0)
    uploaded, window = [], []
    orig_c = c_opt.optimizeFrames
    def counting_c(*a, **k):
        r = orig_c(*a, **k); uploaded.append(c_opt.last_stats["corr_pairs_uploaded"]); window.append(c_opt.last_stats["n_frames"]); return r
    c_opt.optimizeFrames = counting_c
    _, _, fc, _ = run_session(c_opt, n, to_device=up, persistent_frame_cache=True)
    for x, y in zip(fa, fc):
        assert np.array_equal(x.pose_in_model, y.pose_in_model)
    # at least the new frame's w - 1 pairs; more only when the keyframe selection brings two old frames together for the first time
    odd = [(k, u, w) for k, (u, w) in enumerate(zip(uploaded, window)) if u != w - 1]
    assert uploaded[0] == 1 and all(w - 1 <= u <= w * (w - 1) // 2 for u, w in zip(uploaded, window)) and len(odd) <= len(uploaded) // 5, odd


@pytest.mark.gpu
def test_session_with_device_ransac():
    """The whole device-side chain of a tracked frame: RANSAC prunes every pair's matches (the 5 % gross outliers go,
    nothing else), Kabsch initialises the pose, bundle adjustment with the persistent frame cache refines the window."""
    import torch
    from bundletrack_amd.optimizer import OptimizerGpu, Workspace
    from bundletrack_amd.ransac import run_ransac_multi_pair
    dev = torch.device("cuda:0")
    ws = Workspace()
    sizes = []
    def ransac(pairs, matches):
        run_ransac_multi_pair(ws, pairs, matches, n_trials=2000, inlier_dist=0.01, seed=17)
        sizes.extend(len(matches[(a.id, b.id)][0]) for a, b in pairs)
    n = 24
    seq, bundler, frames, errs = run_session(OptimizerGpu(workspace=ws), n, to_device=lambda a: torch.from_numpy(a).to(dev),
                                             persistent_frame_cache=True, ransac=ransac)
    check_session(seq, bundler, frames, errs, n)
    assert min(sizes) >= 280 and max(sizes) <= 290, (min(sizes), max(sizes))     # 300 matches, 15 planted outliers, 1 mm noise vs 10 mm gate
