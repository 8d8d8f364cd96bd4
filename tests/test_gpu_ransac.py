"""HIP correspondence RANSAC (btba_ransac_pairs) against the CPU oracle, through the C ABI."""
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
    res = ransac_multi_pair(ws, [s[0] for s in sets], [s[1] for s in sets], n_trials=n_trials, inlier_dist=0.01, samples=smp, want_trials=True)
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
    """No explicit samples: both sides draw round(u (n-1)) from the same counter hash, so trial for trial they agree."""
    from bundletrack_amd.ransac import ransac_multi_pair
    rng = np.random.default_rng(6)
    sets = [planted(rng, n, f) for n, f in ((40, 0.25), (500, 0.45), (2000, 0.3))]
    res = ransac_multi_pair(ws, [s[0] for s in sets], [s[1] for s in sets], n_trials=2000, inlier_dist=0.01, seed=99, want_trials=True)
    for p, (P, Q, T, mask) in enumerate(sets):
        ref = oracle.ransac_pair(P, Q, 2000, 0.01, seed=99, pair_id=p)
        r = res[p]
        assert np.array_equal(r["counts"] > 0, ref["counts"] > 0)                # same triples skipped
        assert (r["counts"] != ref["counts"]).mean() < 0.02                      # borderline points only
        assert np.array_equal(r["inlier_ids"], np.nonzero(mask)[0]) and np.array_equal(r["inlier_ids"], ref["inlier_ids"])
        e = S.pose_error(r["best_pose"], T)
        assert e[0] < 0.05 and e[1] < 0.005
    again = ransac_multi_pair(ws, [s[0] for s in sets], [s[1] for s in sets], n_trials=2000, inlier_dist=0.01, seed=99, want_trials=True)
    for a, b in zip(res, again):                                                 # deterministic, bit for bit
        assert np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["poses"], b["poses"]) and a["best_trial"] == b["best_trial"]


def test_edge_cases_and_caller_logic(ws):
    from bundletrack_amd import _lib
    from bundletrack_amd.bundler import FrameRef
    from bundletrack_amd.ransac import ransac_multi_pair, run_ransac_multi_pair
    rng = np.random.default_rng(7)
    P, Q, T, mask = planted(rng, 50, 0.2)
    empty = np.zeros((0, 3), np.float32)
    res = ransac_multi_pair(ws, [empty, P[:2], P], [empty, Q[:2], Q], n_trials=300, seed=1)
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
