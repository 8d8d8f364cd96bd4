"""HIP correspondence RANSAC (btba_ransac_pairs / btba_ransac_pairs_ex) through the C ABI: the reference-exact hypotheses against
the reference's own procrustesKernel / evalPoseKernel (oracle/_ref), the Horn hypotheses against the CPU oracle."""
import os

import numpy as np
import pytest

from bundletrack_amd import synthetic as S
from test_oracle_ransac import planted

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ws():
    import torch
    assert torch.cuda.is_available()
    from bundletrack_amd.optimizer import Workspace
    return Workspace()


def test_hypotheses_and_votes_match_oracle(ws, oracle):
    """Same explicit sample triples on both sides: every trial's pose within 1e-4, every trial's inlier count equal
    except for points whose distance sits within 2e-6 m of the gate, the same winner, the same inlier list."""
    from bundletrack_amd.ransac import ransac_multi_pair
    rng = np.random.default_rng(5)
    sets = [planted(rng, n, f) for n, f in ((12, 0.0), (60, 0.3), (300, 0.5), (700, 0.2))]
    n_trials = 512
    smp = np.stack([rng.integers(0, len(s[0]), size=(n_trials, 3)) for s in sets]).astype(np.int32)
    smp[:, 0] = [[0, 0, 1]] * len(sets); smp[:, 1, 1] = -1                      # degenerate triples are skipped
    from bundletrack_amd import _lib
    res = ransac_multi_pair(ws, [s[0] for s in sets], [s[1] for s in sets], n_trials=n_trials, inlier_dist=0.01, samples=smp, want_trials=True, hypothesis=_lib.RANSAC_HORN)
    for p, (P, Q, T, mask) in enumerate(sets):
        ref = oracle.ransac_pair(P, Q, n_trials, 0.01, samples=smp[p])
        r = res[p]
        used = ref["counts"] > 0
        assert np.array_equal(r["counts"] > 0, used)
        dpose = np.abs(r["poses"][used] - ref["poses"][used][:, :3, :]).max()
        assert dpose < 1e-4, dpose
        # counts: recount the oracle's borderline points
        P4 = np.concatenate([P, np.ones((len(P), 1), np.float32)], 1).astype(np.float64)
        for t in np.nonzero(r["counts"] != ref["counts"])[0]:
            dist = np.linalg.norm(Q.astype(np.float64) - P4 @ ref["poses"][t][:3].astype(np.float64).T, axis=1)
            assert abs(int(r["counts"][t]) - int(ref["counts"][t])) <= int((np.abs(dist - 0.01) < 2e-6).sum()), t
        assert r["best_trial"] == ref["best_trial"]
        assert np.array_equal(r["inlier_ids"], ref["inlier_ids"]) and np.array_equal(r["inlier_ids"], np.nonzero(mask)[0])
        assert np.abs(r["best_pose"] - ref["best_pose"]).max() < 1e-4
        assert r["counts"][r["best_trial"]] == len(r["inlier_ids"])               # the vote and the list agree


def test_device_sampling_matches_oracle_draws(ws, oracle):
    """No explicit samples, BTBA_RANSAC_DRAW_HASH: both sides draw round(u (n-1)) from the same counter hash, so trial for trial they agree."""
    from bundletrack_amd.ransac import ransac_multi_pair
    rng = np.random.default_rng(6)
    sets = [planted(rng, n, f) for n, f in ((40, 0.25), (500, 0.45), (2000, 0.3))]
    from bundletrack_amd import _lib
    hyp = _lib.RANSAC_HORN | _lib.RANSAC_DRAW_HASH
    res = ransac_multi_pair(ws, [s[0] for s in sets], [s[1] for s in sets], n_trials=2000, inlier_dist=0.01, seed=99, want_trials=True, hypothesis=hyp)
    for p, (P, Q, T, mask) in enumerate(sets):
        ref = oracle.ransac_pair(P, Q, 2000, 0.01, seed=99, pair_id=p)
        r = res[p]
        assert np.array_equal(r["counts"] > 0, ref["counts"] > 0)                # same triples skipped
        assert (r["counts"] != ref["counts"]).mean() < 0.02                      # borderline points only
        assert np.array_equal(r["inlier_ids"], np.nonzero(mask)[0]) and np.array_equal(r["inlier_ids"], ref["inlier_ids"])
        e = S.pose_error(r["best_pose"], T)
        assert e[0] < 0.05 and e[1] < 0.005
    again = ransac_multi_pair(ws, [s[0] for s in sets], [s[1] for s in sets], n_trials=2000, inlier_dist=0.01, seed=99, want_trials=True, hypothesis=hyp)
    for a, b in zip(res, again):                                                 # deterministic, bit for bit
        assert np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["poses"], b["poses"]) and a["best_trial"] == b["best_trial"]


def test_edge_cases_and_caller_logic(ws):
    from bundletrack_amd import _lib
    from bundletrack_amd.bundler import FrameRef
    from bundletrack_amd.ransac import ransac_multi_pair, run_ransac_multi_pair
    rng = np.random.default_rng(7)
    P, Q, T, mask = planted(rng, 50, 0.2)
    empty = np.zeros((0, 3), np.float32)
    res = ransac_multi_pair(ws, [empty, P[:2], P], [empty, Q[:2], Q], n_trials=300, seed=1, hypothesis=_lib.RANSAC_DRAW_HASH)
    assert res[0]["best_trial"] == -1 and len(res[0]["inlier_ids"]) == 0        # empty pair
    assert res[1]["best_trial"] == -1 and len(res[1]["inlier_ids"]) == 0        # two points: nothing to fit
    assert np.array_equal(res[2]["inlier_ids"], np.nonzero(mask)[0])
    with pytest.raises(ValueError):
        ransac_multi_pair(ws, [P], [Q[:10]])
    with pytest.raises(_lib.BtbaError):
        ransac_multi_pair(ws, [P], [Q], n_trials=0)
    # runRansacMultiPairGPU: camera-frame matches, poses applied on the host, inliers kept, < 5 survivors => emptied
    Ta, Tb = S.orbit_pose(0.3).astype(np.float32), S.orbit_pose(0.1).astype(np.float32)
    fa, fb, fc = FrameRef(id=2, pose_in_model=Ta), FrameRef(id=1, pose_in_model=Tb), FrameRef(id=0, pose_in_model=Tb)
    world = rng.uniform(-0.05, 0.05, size=(80, 3))
    inv = lambda M: np.linalg.inv(M.astype(np.float64))
    pa = (world @ inv(Ta)[:3, :3].T + inv(Ta)[:3, 3]).astype(np.float32)
    pb = (world @ inv(Tb)[:3, :3].T + inv(Tb)[:3, 3]).astype(np.float32)
    pb[:16] += 0.05                                                             # 16 gross outliers
    junk_a, junk_b = rng.uniform(-1, 1, (6, 3)).astype(np.float32), rng.uniform(-1, 1, (6, 3)).astype(np.float32)
    matches = {(2, 1): (pa, pb), (2, 0): (junk_a, junk_b)}
    run_ransac_multi_pair(ws, [(fa, fb), (fa, fc)], matches, n_trials=500, inlier_dist=0.01, seed=3)
    assert len(matches[(2, 1)][0]) == 64 and np.array_equal(matches[(2, 1)][0], pa[16:])
    assert len(matches[(2, 0)][0]) == 0


def test_reference_hypotheses_match_the_reference_kernels(ws):
    """The default hypothesis (BTBA_RANSAC_REFERENCE_SVD) against the reference's OWN procrustesKernel (with the pasted approximate
    3x3 SVD) and evalPoseKernel, compiled for the CPU (oracle/_ref/libbtba_ref_ransac.so), on identical explicit sample triples:
    every trial's pose within 1e-6, every trial's inlier count identical, the same winner, the same inlier list -- on planted
    sets, on 3-point samples with 1 cm noise (where the approximate SVD is furthest from the Kabsch optimum), and with degenerate
    triples in the list."""
    from oracle import reference as R
    if not os.path.exists(R.SO_RANSAC):
        pytest.skip("oracle/_ref/libbtba_ref_ransac.so not built")
    from bundletrack_amd.ransac import ransac_multi_pair
    rng = np.random.default_rng(15)
    sets = [planted(rng, n, f) for n, f in ((12, 0.0), (60, 0.3), (300, 0.5), (700, 0.2))]
    noisy = planted(rng, 300, 0.1)
    noisy = (noisy[0], (noisy[1] + rng.normal(size=noisy[1].shape).astype(np.float32) * 0.01).astype(np.float32), noisy[2], noisy[3])
    sets.append(noisy)
    n_trials = 384
    smp = np.stack([rng.integers(0, len(s[0]), size=(n_trials, 3)) for s in sets]).astype(np.int32)
    smp[:, 0] = [[0, 0, 1]] * len(sets); smp[:, 1, 1] = -1                      # skipped like ransacEstimateModelKernel skips them (:1164-1165)
    res = ransac_multi_pair(ws, [s[0] for s in sets], [s[1] for s in sets], n_trials=n_trials, inlier_dist=0.01, samples=smp, want_trials=True)
    worst = 0.0
    for p, (P, Q, T, mask) in enumerate(sets):
        r = res[p]
        want_cnt = np.zeros(n_trials, np.int64)
        want_pose = np.tile(np.eye(4, dtype=np.float32)[:3], (n_trials, 1, 1))
        for t in range(n_trials):
            i = smp[p, t]
            if len(set(i.tolist())) < 3 or (i < 0).any():
                continue
            ok, pose = R.procrustes(P[i], Q[i])
            if ok:
                want_pose[t] = pose[:3]
                want_cnt[t] = len(R.eval_pose(P, Q, pose, 0.01))
        worst = max(worst, float(np.abs(r["poses"] - want_pose).max()))
        assert np.abs(r["poses"] - want_pose).max() <= 1e-6
        assert np.array_equal(r["counts"].astype(np.int64), want_cnt)
        best = int(np.argmax(want_cnt)) if want_cnt.max() > 0 else -1           # most inliers, lowest trial id among equals
        assert r["best_trial"] == best
        if best >= 0:
            pose4 = np.vstack([want_pose[best], [0, 0, 0, 1]]).astype(np.float32)
            assert np.array_equal(r["inlier_ids"], R.eval_pose(P, Q, pose4, 0.01))
    print(f"reference-exact hypotheses: worst |pose - procrustesKernel| over {len(sets) * n_trials} trials = {worst:.1e}")


def test_device_resident_ransac_equals_the_host_buffer_form(ws):
    """btba_ransac_pairs_ex(device_resident = 1): points and results stay on the GPU; same bits as the host-buffer call."""
    import torch
    from bundletrack_amd.ransac import pack_points, ransac_packed, ransac_packed_device
    rng = np.random.default_rng(16)
    sets = [planted(rng, n, f) for n, f in ((40, 0.25), (300, 0.4), (900, 0.3))]
    a_all, b_all, n_pts = pack_points([s[0] for s in sets], [s[1] for s in sets])
    host = ransac_packed(ws, a_all, b_all, n_pts, n_trials=1000, inlier_dist=0.01, seed=5)
    dev = torch.device("cuda:0")
    ids, n_in, best, pose = ransac_packed_device(ws, torch.from_numpy(a_all).to(dev), torch.from_numpy(b_all).to(dev), n_pts, n_trials=1000, inlier_dist=0.01, seed=5)
    ws.sync()
    ids, n_in, best, pose = ids.cpu().numpy(), n_in.cpu().numpy(), best.cpu().numpy(), pose.cpu().numpy()
    o = 0
    for p, h in enumerate(host):
        assert n_in[p] == len(h["inlier_ids"]) and np.array_equal(ids[o:o + n_in[p]], h["inlier_ids"])
        assert best[p] == h["best_trial"] and np.array_equal(pose[p].reshape(4, 4), h["best_pose"])
        o += int(n_pts[p])


def test_horn_and_reference_hypotheses_pick_the_same_inliers_on_tracker_size_pairs(ws):
    """2 000 trials on tracker-size pairs (300 matches, 5-30 % gross outliers, 1 mm noise): the exact Kabsch hypotheses and the
    reference's approximate-SVD hypotheses end with the same inlier set (the winning trial may differ)."""
    from bundletrack_amd import _lib
    from bundletrack_amd.ransac import ransac_multi_pair
    rng = np.random.default_rng(17)
    sets = []
    for f in (0.05, 0.1, 0.2, 0.3):
        P, Q, T, mask = planted(rng, 300, f)
        Q = (Q + rng.normal(size=Q.shape).astype(np.float32) * 0.001 * mask[:, None]).astype(np.float32)
        sets.append((P, Q, T, mask))
    a = ransac_multi_pair(ws, [s[0] for s in sets], [s[1] for s in sets], n_trials=2000, inlier_dist=0.01, seed=7, hypothesis=_lib.RANSAC_DRAW_HASH)
    b = ransac_multi_pair(ws, [s[0] for s in sets], [s[1] for s in sets], n_trials=2000, inlier_dist=0.01, seed=7, hypothesis=_lib.RANSAC_HORN | _lib.RANSAC_DRAW_HASH)
    for (P, Q, T, mask), ra, rb in zip(sets, a, b):
        assert np.array_equal(ra["inlier_ids"], rb["inlier_ids"]) and np.array_equal(ra["inlier_ids"], np.nonzero(mask)[0])


def test_default_draw_is_the_restated_curand_stream(ws, oracle):
    """samples = NULL, seed = 0 (the drop-in default): trial t of every pair samples round(u (n-1)) from cuRAND's XORWOW stream of
    curand_init(0, t, 0) -- cuda_ransac.cu:1154-1161.  The product builds that stream on the host (btba_xorwow.hpp) and the vote
    kernel reads it as a table; here the same call is repeated with the ORACLE's restatement of the stream (oracle/xorwow.h) passed
    as explicit triples, and with the oracle's whole RANSAC on those triples.  Per-trial poses and counts, winner and inlier lists
    must be identical -- for every pair, including pairs of different sizes in one launch, n_trials not a multiple of the
    workgroup, a second seed, and a workspace whose cached table is first shorter, then for another seed."""
    from bundletrack_amd import _lib
    from bundletrack_amd.ransac import ransac_multi_pair
    rng = np.random.default_rng(23)
    sets = [planted(rng, n, f, noise=0.002) for n, f in ((3, 0.0), (7, 0.0), (40, 0.25), (300, 0.3), (301, 0.5), (2000, 0.3))]
    A, B = [s[0] for s in sets], [s[1] for s in sets]
    for n_trials, seed in ((100, 0), (777, 0), (2000, 0), (500, 12345), (2000, 0)):
        dflt = ransac_multi_pair(ws, A, B, n_trials=n_trials, inlier_dist=0.01, seed=seed, want_trials=True)
        smp = np.stack([oracle.ransac_reference_samples(n_trials, len(P), seed) for P in A])
        expl = ransac_multi_pair(ws, A, B, n_trials=n_trials, inlier_dist=0.01, samples=smp, want_trials=True)
        for p, (d, e) in enumerate(zip(dflt, expl)):
            assert np.array_equal(d["counts"], e["counts"]) and np.array_equal(d["poses"].view(np.uint32), e["poses"].view(np.uint32)), (n_trials, seed, p)
            assert d["best_trial"] == e["best_trial"] and np.array_equal(d["inlier_ids"], e["inlier_ids"])
            ref = oracle.ransac_pair(A[p], B[p], n_trials, 0.01, samples=smp[p], hypothesis=0)
            assert np.array_equal(d["counts"] > 0, ref["counts"] > 0)
            assert (d["counts"] != ref["counts"]).mean() < 0.02                  # points within an ulp of the gate only
            if np.array_equal(d["counts"], ref["counts"]):
                assert d["best_trial"] == ref["best_trial"] and np.array_equal(d["inlier_ids"], ref["inlier_ids"])
    # the stream is the same for every pair: two pairs with the same number of points draw the same triples
    same = ransac_multi_pair(ws, [A[3], A[3]], [B[3], B[3]], n_trials=300, inlier_dist=0.01, want_trials=True)
    assert np.array_equal(same[0]["counts"], same[1]["counts"]) and np.array_equal(same[0]["poses"], same[1]["poses"])
    hashed = ransac_multi_pair(ws, [A[3], A[3]], [B[3], B[3]], n_trials=300, inlier_dist=0.01, want_trials=True, hypothesis=_lib.RANSAC_DRAW_HASH)
    assert not np.array_equal(hashed[0]["counts"], hashed[1]["counts"])            # the counter hash gives every pair its own triples


def test_default_call_against_the_references_whole_ransac(ws, oracle):
    """btba_ransac_pairs' default (samples NULL, seed 0) against ransacMultiPairGPU itself -- the reference's three kernels and host
    launcher compiled from /root/reference for the CPU (oracle/_ref/libbtba_ref_ransac.so; cuRAND behind it is oracle/xorwow.h) -- end to
    end, several pairs in one call on both sides: the same inlier lists.  Where several trials tie for the most inliers the emulated
    findBestTrial keeps the last of them (a race on a real GPU), the product the first; the product's per-trial counts name the tied
    trials and the reference's list must then be the product's list for the last one."""
    from oracle import reference as R
    if not os.path.exists(R.SO_RANSAC) or not hasattr(R.lib_ransac(), "ref_ransac_multi_pair"):
        pytest.skip("oracle/_ref/libbtba_ref_ransac.so (with ransacMultiPairGPU) not built")
    from bundletrack_amd.ransac import ransac_multi_pair, reference_samples
    rng = np.random.default_rng(29)
    sets = [planted(rng, n, f, noise=0.002) for n, f in ((5, 0.0), (40, 0.3), (300, 0.5), (301, 0.5), (700, 0.2), (1500, 0.4))]
    A, B = [s[0] for s in sets], [s[1] for s in sets]
    n_trials = 600
    want = R.ransac_multi_pair(A, B, n_trials, 0.01)
    got = ransac_multi_pair(ws, A, B, n_trials=n_trials, inlier_dist=0.01, seed=0, want_trials=True)
    ties = 0
    for p, (g, w) in enumerate(zip(got, want)):
        counts = g["counts"]
        if counts.max() == 0:
            assert len(w) == 0 and g["best_trial"] == -1
            continue
        tied = np.nonzero(counts == counts.max())[0]
        assert g["best_trial"] == tied[0] and len(w) == counts.max(), p
        if len(tied) == 1:
            assert np.array_equal(g["inlier_ids"], w), p
        else:
            ties += 1
            smp = reference_samples(n_trials, len(A[p]))[tied[-1]][None, None, :]
            last = ransac_multi_pair(ws, [A[p]], [B[p]], n_trials=1, inlier_dist=0.01, samples=smp)[0]
            assert np.array_equal(last["inlier_ids"], w), p
    print(f"HIP default call vs the reference's ransacMultiPairGPU: {len(sets)} pairs identical ({ties} through the tie rule)")
