"""Correspondence RANSAC timing: all 105 frame pairs of a K=15 window, 2 000 matches per pair, 2 000 trials (the
reference's ransac.max_iter) in ONE call, against the CPU oracle on a sample of pairs.  GPU box only."""
import json, os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(_TESTS)); sys.path.insert(0, _TESTS)
import numpy as np, torch
from bundletrack_amd.optimizer import Workspace
from bundletrack_amd.ransac import pack_points, ransac_packed
from oracle import oracle as O
from test_oracle_ransac import planted


def main():
    rng = np.random.default_rng(0)
    ws = Workspace()
    rows = []
    for n_pairs, n_pts in ((1, 300), (10, 300), (105, 300), (105, 2000)):
        sets = [planted(rng, n_pts, 0.3) for _ in range(n_pairs)]
        A, B = [s[0] for s in sets], [s[1] for s in sets]
        a_all, b_all, npts = pack_points(A, B)          # host marshalling (what FeatureManager.cpp:680-705 does) is not timed
        for _ in range(3): res = ransac_packed(ws, a_all, b_all, npts, n_trials=2000, inlier_dist=0.01, seed=1)
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps): res = ransac_packed(ws, a_all, b_all, npts, n_trials=2000, inlier_dist=0.01, seed=1)
        dt = (time.perf_counter() - t0) / reps
        for _ in range(2): ransac_packed(ws, a_all, b_all, npts, n_trials=8, inlier_dist=0.01, seed=1)
        t0 = time.perf_counter()
        for _ in range(reps): ransac_packed(ws, a_all, b_all, npts, n_trials=8, inlier_dist=0.01, seed=1)
        floor = (time.perf_counter() - t0) / reps           # same call with 8 trials: H2D + D2H + launch floor
        ok = all(np.array_equal(r["inlier_ids"], np.nonzero(s[3])[0]) for r, s in zip(res, sets))
        t0 = time.perf_counter()
        ns = min(n_pairs, 2)
        for p in range(ns): O.ransac_pair(A[p], B[p], 2000, 0.01, seed=1, pair_id=p)
        cpu = (time.perf_counter() - t0) / ns
        rows.append(dict(n_pairs=n_pairs, n_pts=n_pts, n_trials=2000, gpu_ms_per_call=round(dt * 1e3, 3), ms_per_call_with_8_trials=round(floor * 1e3, 3),
                         ghyp_points_per_s_whole_call=round(n_pairs * 2000.0 * n_pts / dt / 1e9, 1),
                         ghyp_points_per_s_above_floor=round(n_pairs * 2000.0 * n_pts / max(dt - floor, 1e-9) / 1e9, 1),
                         cpu_oracle_ms_per_pair=round(cpu * 1e3, 2), speedup_vs_1_cpu_thread=round(cpu * n_pairs / dt, 1), planted_inliers_recovered=bool(ok)))
        print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
