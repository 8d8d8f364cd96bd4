"""BASELINE.json's configurations at full size on the GPU: parity against the oracle where it finishes in
seconds, plus size-independent properties (determinism, finiteness, objective decrease, instance independence)."""
import os

import numpy as np
import pytest

from bundletrack_amd import _lib, synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    from bundletrack_amd.optimizer import BatchSolver, Workspace
    class G:
        pass
    g = G()
    g.torch, g.dev, g.ws, g.BatchSolver = torch, torch.device("cuda:0"), Workspace(), BatchSolver
    return g


def run_gpu(g, pbs, layout="zn", **params):
    """Batched solve with a trace.  layout "zn" = the compact (gated z, normal) cache and the pinhole sweep -- what bench.py and
    btba_optimize_frames run; "float4" = the reference-layout caches (btba_solve_batch)."""
    caches = [S.analytic_cache(pb) for pb in pbs]
    bs = g.BatchSolver(g.ws, **params)
    N = pbs[0].n_frames
    corr, offs, mx = bs.pack_correspondences([pb.corr for pb in pbs], N)
    B = len(pbs)
    corr_d = g.torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(g.dev)
    offs_d = g.torch.from_numpy(offs.astype(np.int32)).to(g.dev)
    poses_d = g.torch.from_numpy(np.stack([pb.poses_init for pb in pbs])).to(g.dev)
    if layout == "zn":
        zn_d = g.torch.from_numpy(np.stack([S.compact_cache(pb) for pb in pbs])).to(g.dev)
        tr = bs.solve_zn(zn_d, pbs[0].H, pbs[0].W, pbs[0].K, corr_d, offs_d, mx, poses_d, trace=True)
    else:
        cam_d = g.torch.from_numpy(np.stack([c[0] for c in caches])).to(g.dev)
        nrm_d = g.torch.from_numpy(np.stack([c[1] for c in caches])).to(g.dev)
        tr = bs.solve(cam_d, nrm_d, caches[0][2], corr_d, offs_d, mx, poses_d, trace=True)
    tv = bs.trace_view(tr)
    return poses_d.cpu().numpy(), tv, caches


def sparse_objective(pb, poses, delta=0.005):
    T = np.asarray(poses, np.float64)
    wi = np.einsum("nab,nb->na", T[pb.corr["imgIdx_i"], :3, :3], pb.corr["pos_i"].astype(np.float64)) + T[pb.corr["imgIdx_i"], :3, 3]
    wj = np.einsum("nab,nb->na", T[pb.corr["imgIdx_j"], :3, :3], pb.corr["pos_j"].astype(np.float64)) + T[pb.corr["imgIdx_j"], :3, 3]
    e = np.sqrt(((wi - wj) ** 2).sum(1))
    return float(np.where(e <= delta, e * e, 2 * delta * e - delta * delta).sum())


@pytest.mark.parametrize("cfg", [
    dict(name="c2", K=10, m=1000, wd=0.0, config=2),      # K=10, 1k corr/pair, feature residuals only
    dict(name="c3", K=15, m=2000, wd=1.0, config=3),      # K=15, 2k corr/pair, feature + dense ICP + Huber (the headline)
    dict(name="c4", K=30, m=4000, wd=1.0, config=4),      # K=30, 4k corr/pair, 60-keyframe pool pruned to 30 by greedy-rot
    dict(name="c3-float4-cache", K=15, m=2000, wd=1.0, config=3, layout="float4"),      # the reference-layout caches (btba_solve_batch)
], ids=lambda c: c["name"])
def test_baseline_configs_match_oracle(gpu, oracle, cfg):
    seed = S.config_seed(cfg["config"])
    angles = S.pruned_pool_angles(60, 30, seed) if cfg["name"] == "c4" else None
    pb = S.make_problem(cfg["K"], cfg["m"], seed, background=True, full_res=False, angles=angles)
    out, tv, caches = run_gpu(gpu, [pb], layout=cfg.get("layout", "zn"), weight_dense_depth=cfg["wd"])
    campos, normals, intr = caches[0]
    ref = oracle.solve(campos, normals, intr, pb.corr, pb.poses_init, params=oracle.default_params(weight_dense_depth=cfg["wd"]))
    worst_r = worst_t = 0.0
    for it in range(7):
        for k in range(pb.n_frames):
            r, t = S.pose_error(tv.T_after[0, it, k], ref.T_after[it, k])
            worst_r, worst_t = max(worst_r, r), max(worst_t, t)
    print(f"{cfg['name']}: worst per-iterate diff rot {worst_r:.3e} rad trans {worst_t:.3e} m")
    assert worst_r < 1e-4 and worst_t < 1e-4
    # decisions: the epsilon guards of PCGStep_Kernel2/3 take the same branch at every PCG step of every
    # iterate, and the accept tests agree up to the handful of pixels that sit on a threshold (v_rcp / v_rsq vs IEEE division)
    hp, op = tv.pcg_scalars[0], ref.pcg_scalars
    assert np.array_equal(hp[..., 1] == 0, op[..., 1] == 0) and np.array_equal(hp[..., 3] == 0, op[..., 3] == 0), "alpha / beta guard traces differ"
    if cfg["wd"] > 0:
        dc = np.abs(np.rint(tv.dense_pair[0][..., 27]).astype(np.int64) - ref.dense_count.astype(np.int64))
        print(f"{cfg['name']}: accepted-pixel counts differ by at most {dc.max()} per pair ({(dc > 0).mean():.1%} of (iterate, pair) cells)")
        assert dc.max() <= 8, dc.max()
    assert np.isfinite(out).all()
    assert sparse_objective(pb, out[0]) < 0.5 * sparse_objective(pb, pb.poses_init)
    e0 = max(S.pose_error(pb.poses_init[k], pb.poses_gt[k])[0] for k in range(pb.n_frames))
    e1 = max(S.pose_error(out[0, k], pb.poses_gt[k])[0] for k in range(pb.n_frames))
    assert e1 < e0


def test_c3_realistic_mask_matches_oracle(gpu, oracle):
    pb = S.make_problem(15, 2000, S.config_seed(3, 900), background=False, full_res=False)
    out, tv, caches = run_gpu(gpu, [pb])
    ref = oracle.solve(caches[0][0], caches[0][1], caches[0][2], pb.corr, pb.poses_init)
    for it in range(7):
        for k in range(15):
            r, t = S.pose_error(tv.T_after[0, it, k], ref.T_after[it, k])
            assert r < 1e-4 and t < 1e-4, (it, k, r, t)


def test_dead_block_skip_is_exact(gpu):
    """The dense sweep drops 8 x 8 blocks whose frustum segment provably projects outside the target image before walking them
    (k_block_ranges + the hull test in dense_block_pinhole).  It must never drop a block that holds an accepted pixel: with the
    skip disabled (BTBA_OPT_BLOCK_SKIP = 0, every block walked) the accepted-pixel counts of every dense pair are IDENTICAL -- at the
    first linearisation on every window, and at all seven iterates on the well-posed and the masked windows --, the sums agree to fp32
    grouping, and the solves stay within round-off of each other.
    Eight windows, among them ones with poses far from the truth (large relative motion: most blocks dead) and a masked one."""
    pbs = [S.make_problem(15, 2000, S.config_seed(5, 40 + b), background=(b != 7), full_res=False) for b in range(8)]
    rng = np.random.default_rng(5)
    for b in (4, 5, 6):                                  # poses far off: rotations up to ~0.5 rad, translations up to 0.2 m about the scene
        for k in range(1, 15):
            pbs[b].poses_init[k] = (S.se3_exp(rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.2, 0.2, 3)) @ pbs[b].poses_init[k].astype(np.float64)).astype(np.float32)
    out_a, tv_a, _ = run_gpu(gpu, pbs)
    gpu.ws.set_option(_lib.OPT_BLOCK_SKIP, 0)
    try:
        out_b, tv_b, _ = run_gpu(gpu, pbs)
    finally:
        gpu.ws.set_option(_lib.OPT_BLOCK_SKIP, 1)
    ca, cb = tv_a.dense_pair[:, 0, :, 27], tv_b.dense_pair[:, 0, :, 27]
    assert np.array_equal(ca, cb)
    assert ca.sum() > 0
    # ... and at EVERY linearisation point of the solve: after the first iterate the two runs no longer walk bit-identical poses (the live
    # list decides which wave sums which block, so their sums are grouped differently and a borderline pixel may flip on the last bit),
    # hence both variants are RESTARTED, one Gauss-Newton iteration each, from the poses run A had after iterates 0 .. 5: at identical
    # poses the skip may not change a single accepted-pixel count, on any of the eight windows.
    bs1 = gpu.BatchSolver(gpu.ws, n_gn_iters=1)
    N = pbs[0].n_frames
    corr, offs, mx = bs1.pack_correspondences([pb.corr for pb in pbs], N)
    zn_r = gpu.torch.from_numpy(np.stack([S.compact_cache(pb) for pb in pbs])).to(gpu.dev)
    corr_r = gpu.torch.from_numpy(corr.view(np.uint8).reshape(len(pbs), -1, 32)).to(gpu.dev)
    offs_r = gpu.torch.from_numpy(offs.astype(np.int32)).to(gpu.dev)
    n_checked = 0
    for it in range(tv_a.T_after.shape[1] - 1):
        start = np.ascontiguousarray(tv_a.T_after[:, it]).astype(np.float32)
        counts = []
        for skip in (1, 0):
            gpu.ws.set_option(_lib.OPT_BLOCK_SKIP, skip)
            try:
                poses_r = gpu.torch.from_numpy(start.copy()).to(gpu.dev)
                tvr = bs1.trace_view(bs1.solve_zn(zn_r, pbs[0].H, pbs[0].W, pbs[0].K, corr_r, offs_r, mx, poses_r, trace=True))
            finally:
                gpu.ws.set_option(_lib.OPT_BLOCK_SKIP, 1)
            counts.append(tvr.dense_pair[:, 0, :, 27].copy())
        assert np.array_equal(counts[0], counts[1]), (it, np.argwhere(counts[0] != counts[1])[:5])
        n_checked += counts[0].size
    print(f"dead-block skip: {n_checked} (window, iterate, pair) accepted-pixel counts identical with and without it")
    print(f"accepted pixels per pair: min {ca.min():.0f} max {ca.max():.0f}; empty pairs {(ca == 0).sum()} of {ca.size}")
    ref = np.abs(tv_b.dense_pair[:, 0]).max(axis=(1, 2), keepdims=True)
    assert (np.abs(tv_a.dense_pair[:, 0] - tv_b.dense_pair[:, 0]) <= 4e-6 * ref).all()
    for b in range(4):                                   # well-posed windows: the whole solve within round-off
        for k in range(15):
            r, t = S.pose_error(out_a[b, k], out_b[b, k])
            assert r < 5e-5 and t < 5e-5, (b, k, r, t)
    # block ranges built once with the cache (btba_zn_block_ranges) and handed to the solve: the same bits as computing them per solve
    bs = gpu.BatchSolver(gpu.ws)
    corr, offs, mx = bs.pack_correspondences([pb.corr for pb in pbs], 15)
    zn_d = gpu.torch.from_numpy(np.stack([S.compact_cache(pb) for pb in pbs])).to(gpu.dev)
    corr_d = gpu.torch.from_numpy(corr.view(np.uint8).reshape(len(pbs), -1, 32)).to(gpu.dev)
    offs_d = gpu.torch.from_numpy(offs.astype(np.int32)).to(gpu.dev)
    poses_d = gpu.torch.from_numpy(np.stack([pb.poses_init for pb in pbs])).to(gpu.dev)
    aux = bs.cache_aux(zn_d, valid_lists=True)
    bs.solve_zn(zn_d, pbs[0].H, pbs[0].W, pbs[0].K, corr_d, offs_d, mx, poses_d, aux=aux)
    assert np.array_equal(poses_d.cpu().numpy(), out_a)
    rng = aux["block_ranges"].cpu().numpy().reshape(len(pbs), 15, -1, 2)
    z = np.stack([S.compact_cache(pb) for pb in pbs])[..., 0]
    zb = z.reshape(len(pbs), 15, z.shape[2] // 8, 8, z.shape[3] // 8, 8).transpose(0, 1, 2, 4, 3, 5).reshape(len(pbs), 15, -1, 64)
    lo = np.where(zb > 0, zb, np.inf).min(-1); hi = np.where(zb > 0, zb, -np.inf).max(-1)
    assert np.array_equal(rng[..., 0], lo) and np.array_equal(rng[..., 1], hi)
    # ... and the valid-pixel lists: supplied lists == lists built inside the solve, and they are the ascending valid pixels
    bl = gpu.BatchSolver(gpu.ws, flags=_lib.FLAG_COMPACTION)
    pa, pb_ = gpu.torch.from_numpy(np.stack([pb.poses_init for pb in pbs])).to(gpu.dev), gpu.torch.from_numpy(np.stack([pb.poses_init for pb in pbs])).to(gpu.dev)
    bl.solve_zn(zn_d, pbs[0].H, pbs[0].W, pbs[0].K, corr_d, offs_d, mx, pa)
    bl.solve_zn(zn_d, pbs[0].H, pbs[0].W, pbs[0].K, corr_d, offs_d, mx, pb_, aux=aux)
    assert np.array_equal(pa.cpu().numpy(), pb_.cpu().numpy())
    cnt = aux["valid_counts"].cpu().numpy(); lst = aux["valid_lists"].cpu().numpy()
    zf = z.reshape(len(pbs) * 15, -1)
    assert np.array_equal(cnt, (zf >= 0.1).sum(1))
    for f in (0, 17, len(cnt) - 1):
        assert np.array_equal(lst[f, :cnt[f]], np.nonzero(zf[f] >= 0.1)[0])


def test_c5_shape_batch_properties(gpu):
    """32 instances of the c3 shape (config 5's per-GPU share): run-to-run bit-identical, every instance equal
    to its own single-instance run up to fp32 noise, all finite, all improved."""
    pbs32 = [S.make_problem(15, 2000, S.config_seed(5, b), background=True, full_res=False) for b in range(32)]      # 32 distinct seeds
    pbs = pbs32[:4]
    out, _, _ = run_gpu(gpu, pbs32)
    out2, _, _ = run_gpu(gpu, pbs32)
    assert np.array_equal(out, out2) and np.isfinite(out).all()
    from bundletrack_amd import _lib
    out3, _, _ = run_gpu(gpu, pbs32, flags=_lib.FLAG_OVERLAP)         # two-stream half-batch pipeline: same bits as one stream
    assert np.array_equal(out, out3)
    out4, _, _ = run_gpu(gpu, pbs32, flags=_lib.FLAG_NO_FUSE)         # sparse and dense sweeps as separate launches: the same functions compiled into other kernels
    assert max(max(S.pose_error(out4[b, k], out[b, k])) for b in range(32) for k in range(15)) < 5e-5
    out5, _, _ = run_gpu(gpu, [pbs32[(b + 7) % 32] for b in range(32)])
    for b in range(32):
        assert np.array_equal(out5[b], out[(b + 7) % 32])    # same instance data -> same bits wherever it sits in the grid
    for b in range(4):
        single, _, _ = run_gpu(gpu, [pbs[b]])
        for k in range(15):
            r, t = S.pose_error(single[0, k], out[b, k])
            assert r < 5e-5 and t < 5e-5      # different tile / chunk counts = different fp32 grouping of the sums, amplified over 7 iterates (measured <= 2.2e-5; the parity bar is 1e-4)
        assert sparse_objective(pbs[b], out[b]) < 0.5 * sparse_objective(pbs[b], pbs[b].poses_init)


def test_one_tracker_window_stays_inside_its_recorded_budget():
    """The operating point of the reference's tracker -- ONE object-masked window per call (Bundler.cpp:350-351) -- watched by a gate: the device time of the
    solve inside btba_optimize_frames_keyed (btba_stats.ms_solve, median of 30 steady-state calls).  Round 4's verdict: this number regressed unnoticed while
    the batched headline improved.  A correctness suite must not fail on another GPU, a loaded box or under a profiler (round 5's advisor), so the gate that
    always runs is a RATIO measured in this process: the default path must not be slower than the same call with rounds 1-4's system solve
    (BTBA_OPT_SOLVE_SMALL = 0; recorded 0.134 against 0.209 ms).  The ABSOLUTE budget (tests/golden/tracker_budget.json + 10 %) is checked only on request
    (BTBA_PERF_GATE=1: the builder's campaign on the recording box class)."""
    import json
    import torch
    from bundletrack_amd.optimizer import OptimizerGpu, Workspace
    budget = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tracker_budget.json")))
    dev = torch.device("cuda:0")
    K, m = 15, 2000
    pb = S.make_problem(K, m, seed=S.config_seed(3, 0) + K, background=False)
    depths = [torch.from_numpy(pb.depth[k]).to(dev) for k in range(K)]
    normals = [torch.from_numpy(pb.normals[k]).to(dev) for k in range(K)]

    def median_ms_solve(solve_small):
        ws = Workspace()
        ws.set_option(_lib.OPT_SOLVE_SMALL, solve_small)
        opt = OptimizerGpu(workspace=ws)
        ms = []
        for rep in range(40):
            poses = pb.poses_init.copy()
            opt.optimizeFrames(pb.corr, pb.n_match_per_pair, K, pb.H, pb.W, depths, None, normals, poses, pb.K, frame_keys=list(range(K - 1)) + [1000 + rep])
            if rep >= 10:
                ms.append(opt.last_stats["ms_solve"])
                assert opt.last_stats["cache_frames_built"] == 1
        return float(np.median(ms))
    med, legacy = median_ms_solve(1), median_ms_solve(0)
    print(f"one masked c3 window: ms_solve median {med:.4f} (rounds 1-4's system solve in the same process: {legacy:.4f}; recorded budget {budget['ms_solve']} + {100 * budget['tolerance']:.0f} %)")
    assert med <= 1.02 * legacy, (med, legacy)
    if os.environ.get("BTBA_PERF_GATE") == "1":
        assert med <= budget["ms_solve"] * (1.0 + budget["tolerance"]), (med, budget)


def test_large_caches_keep_the_block_walk_in_chip_filling_batches(gpu):
    """A 320 x 240 cache (image_downscale 2) has 1 200 8 x 8 blocks; the dead-block test handles bands of up to 1 024.  Once a batch fills the chip the tile
    policy asks for ONE tile per pair (measured on 160 x 120 caches) -- which would drop the hull-culled block walk for row strips here (round 4's advisor
    finding).  pick_tiles keeps two tiles in that case: the walk's counter must move, and the poses must equal the same instance solved alone."""
    pb = S.make_problem(4, 60, seed=77, background=True, full_res=False, downscale=2)
    zn1 = S.compact_cache(pb)
    assert zn1.shape[1:3] == (240, 320)
    B = 256                                         # 256 instances x 6 pairs = 1 536 pair-instances: the one-tile regime
    bs = gpu.BatchSolver(gpu.ws, image_downscale=2.0)
    corr, offs, mx = bs.pack_correspondences([pb.corr], 4)
    zn_d = gpu.torch.from_numpy(zn1[None]).to(gpu.dev).repeat(B, 1, 1, 1, 1).contiguous()
    corr_d = gpu.torch.from_numpy(corr.view(np.uint8).reshape(1, -1, 32)).to(gpu.dev).repeat(B, 1, 1).contiguous()
    offs_d = gpu.torch.from_numpy(offs.astype(np.int32)).to(gpu.dev).repeat(B, 1).contiguous()
    poses_d = gpu.torch.from_numpy(pb.poses_init[None].astype(np.float32)).to(gpu.dev).repeat(B, 1, 1, 1).contiguous()
    gpu.ws.set_option(_lib.OPT_COUNT_LIVE, 1)
    try:
        bs.solve_zn(zn_d, pb.H, pb.W, pb.K, corr_d, offs_d, mx, poses_d)
        walked = gpu.ws.live_blocks()
    finally:
        gpu.ws.set_option(_lib.OPT_COUNT_LIVE, 0)
    st = gpu.ws.collect_stats()
    assert st["dense_tiles"] == 2, st
    assert walked > 0, "the block walk was dropped"
    assert walked <= 7 * B * 6 * 1200
    out = poses_d.cpu().numpy()
    assert np.isfinite(out).all() and all(np.array_equal(out[0], out[b]) for b in (1, B // 2, B - 1))
    one = gpu.torch.from_numpy(pb.poses_init[None].astype(np.float32)).to(gpu.dev)
    bs1 = gpu.BatchSolver(gpu.ws, image_downscale=2.0)
    bs1.solve_zn(zn_d[:1], pb.H, pb.W, pb.K, corr_d[:1], offs_d[:1], mx, one)
    gpu.ws.sync()
    worst = max(max(S.pose_error(out[0, k], one.cpu().numpy()[0, k])) for k in range(4))
    print(f"320x240 cache, 256 instances at two tiles vs one instance at its own tile count: worst pose difference {worst:.2e}")
    assert worst < 1e-3, worst                      # (other tile / chunk counts = another summation order on a four-frame 100 %-valid window: the class held to 1e-3, test_gpu_parity.py)
