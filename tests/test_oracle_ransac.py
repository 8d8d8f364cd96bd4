"""Pins for the RANSAC oracle (oracle/btba_oracle_ransac.c).  Self-derived: the reference has no vectors for this
step, its sample generator (cuRAND) and its approximate 3x3 SVD (McAdams) are third-party code absent from the
checkout -- see the oracle's header."""
import os

import numpy as np
import pytest

from bundletrack_amd import synthetic as S


def kabsch64(P, Q):
    """Textbook Kabsch in float64 (what Utils::solveRigidTransformBetweenPoints and procrustesKernel both compute)."""
    P, Q = np.asarray(P, np.float64), np.asarray(Q, np.float64)
    mp, mq = P.mean(0), Q.mean(0)
    U, s, Vt = np.linalg.svd((P - mp).T @ (Q - mq))
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    R = Vt.T @ np.diag([1, 1, d]) @ U.T
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = mq - R @ mp
    return T, s, d


def planted(rng, n, outlier_frac, noise=0.0005, out_lo=0.03, out_hi=0.08):
    P = rng.uniform(-0.08, 0.08, size=(n, 3))
    T = S.se3_exp(rng.uniform(-0.4, 0.4, 3), rng.uniform(-0.05, 0.05, 3))
    Q = P @ T[:3, :3].T + T[:3, 3] + rng.normal(scale=noise, size=(n, 3))
    out = rng.choice(n, int(round(outlier_frac * n)), replace=False)
    dirs = rng.normal(size=(len(out), 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    Q[out] += dirs * rng.uniform(out_lo, out_hi, size=(len(out), 1))
    mask = np.ones(n, bool); mask[out] = False
    return P.astype(np.float32), Q.astype(np.float32), T, mask


def test_procrustes_matches_kabsch(oracle):
    rng = np.random.default_rng(1)
    for n in (3, 3, 3, 4, 10, 200):
        for _ in range(20):
            P, Q, T, _ = planted(rng, n, 0.0, noise=0.002)
            ok, pose, gap = oracle.procrustes(P, Q)
            ref, s, d = kabsch64(P, Q)
            assert ok and np.abs(pose - ref).max() < 2e-5
            assert abs(gap - (s[1] + d * s[2]) / s[0]) < 1e-3
            assert abs(np.linalg.det(pose[:3, :3].astype(np.float64)) - 1) < 1e-5
    # mirrored target: the optimum is still a proper rotation (reflection repaired by flipping V's last column)
    P, Q, _, _ = planted(rng, 50, 0.0)
    Qm = Q * np.array([1, 1, -1], np.float32)
    ok, pose, _ = oracle.procrustes(P, Qm)
    ref, _, d = kabsch64(P, Qm)
    assert ok and d < 0 and np.abs(pose - ref).max() < 2e-5 and np.linalg.det(pose[:3, :3].astype(np.float64)) > 0.999
    # collinear sample: gap ~ 0 -> callers skip it
    line = np.outer(np.array([0.0, 0.01, 0.03]), np.array([1.0, 2.0, -1.0])).astype(np.float32)
    ok, pose, gap = oracle.procrustes(line, line + 0.01)
    assert gap < 1e-4


def test_draw_distribution_and_determinism(oracle):
    n = 17
    d = np.array([[oracle.ransac_draw(7, 3, t, k, n) for k in range(3)] for t in range(4000)])
    assert d.min() == 0 and d.max() == n - 1
    h = np.bincount(d.ravel(), minlength=n) / d.size
    assert abs(h[0] - 0.5 / (n - 1)) < 0.01 and abs(h[-1] - 0.5 / (n - 1)) < 0.01       # round(u (n-1)): end points get half weight
    assert np.abs(h[1:-1] - 1.0 / (n - 1)).max() < 0.015
    assert oracle.ransac_draw(7, 3, 5, 1, n) == oracle.ransac_draw(7, 3, 5, 1, n)
    assert len({oracle.ransac_draw(s, 0, 0, 0, 1000) for s in range(50)}) > 40


def test_ransac_recovers_planted_inliers(oracle):
    rng = np.random.default_rng(2)
    for n, frac in ((40, 0.2), (300, 0.4), (1000, 0.6)):
        P, Q, T, mask = planted(rng, n, frac)
        r = oracle.ransac_pair(P, Q, 2000, 0.01, seed=11, pair_id=0)
        assert r["best_trial"] >= 0
        assert np.array_equal(r["inlier_ids"], np.nonzero(mask)[0])              # 0.5 mm noise vs 10 mm gate vs >= 30 mm outliers
        assert r["counts"].max() == mask.sum() and r["counts"][r["best_trial"]] == mask.sum()
        assert r["best_trial"] == int(np.argmax(r["counts"]))                    # lowest trial id among the maxima
        e = S.pose_error(r["best_pose"], T)
        assert e[0] < 0.05 and e[1] < 0.005                                      # a 3-point hypothesis, not the refined pose
    # explicit samples (the reference's rand_list path): degenerate triples are skipped
    P, Q, T, mask = planted(rng, 30, 0.0)
    smp = np.array([[0, 0, 1], [2, -1, 3], [4, 5, 6], [7, 8, 9]], np.int32)
    r = oracle.ransac_pair(P, Q, 4, 0.01, samples=smp)
    assert list(r["counts"][:2]) == [0, 0] and r["best_trial"] == 2 and len(r["inlier_ids"]) == 30
    # nothing to fit
    r = oracle.ransac_pair(P[:2], Q[:2], 50, 0.01, seed=3)
    assert r["best_trial"] == -1 and len(r["inlier_ids"]) == 0


def test_rsqrt_is_correctly_rounded():
    """btba_svd3.hpp::rsqrt_rn (the product's __frsqrt_rn) returns THE float nearest to x^-1/2: checked with exact integer arithmetic -- for
    a float r with neighbours r-, r+ the claim is  m_lo^2 x < 1 < m_hi^2 x  at the two midpoints, evaluated in Python integers -- on
    two million random inputs over 60 binades, the perfect squares, and inputs constructed to sit next to a midpoint (where the plain
    (float)(1 / sqrt((double) x)) of round 2 can round the wrong way)."""
    import ctypes as C, subprocess
    from fractions import Fraction
    here = os.path.dirname(os.path.abspath(__file__))
    so, src_cpp = os.path.join(here, "cpp", "libsvd3_host.so"), os.path.join(here, "cpp", "svd3_host.cpp")
    hdr = os.path.join(os.path.dirname(here), "bundletrack_amd", "csrc", "btba_svd3.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src_cpp), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unknown-pragmas", "-o", so, src_cpp])
    f = C.CDLL(so).rsqrt_rn_host
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; f.restype = None
    rng = np.random.default_rng(11)
    x = np.concatenate([
        (rng.uniform(1.0, 4.0, 2_000_000) * 2.0 ** rng.integers(-30, 30, 2_000_000)).astype(np.float32),
        (np.arange(1, 4097, dtype=np.float32) ** 2),
        # x = nearest float to 1 / m^2 for float midpoints m: x^-1/2 lies within ~1e-8 relative of m, the hard neighbourhood
        (1.0 / ((np.float32(1.0) + np.arange(1, 200001, 7, dtype=np.float64) * 2.0 ** -23 + 2.0 ** -24) ** 2)).astype(np.float32),
    ])
    out = np.zeros_like(x)
    f(x.ctypes.data, len(x), out.ctypes.data)
    # vectorised screen in extended precision, then the exact check on the closest calls
    ld = np.longdouble
    y = 1.0 / np.sqrt(x.astype(ld))
    up, dn = np.nextafter(out, np.float32(np.inf)), np.nextafter(out, np.float32(0))
    m_hi, m_lo = (out.astype(ld) + up.astype(ld)) / 2, (out.astype(ld) + dn.astype(ld)) / 2
    assert (y < m_hi).all() and (y > m_lo).all()
    margin = np.minimum((m_hi - y) / y, (y - m_lo) / y).astype(np.float64)
    for k in np.argsort(margin)[:2000]:
        X = Fraction(float(x[k]))
        mh, ml = (Fraction(float(out[k])) + Fraction(float(up[k]))) / 2, (Fraction(float(out[k])) + Fraction(float(dn[k]))) / 2
        assert ml * ml * X < 1 < mh * mh * X, (float(x[k]), float(out[k]))
    naive = (1.0 / np.sqrt(x.astype(np.float64))).astype(np.float32)
    print(f"rsqrt_rn: {len(x)} inputs correctly rounded; the plain double-then-float form differs on {(naive != out).sum()} of them; smallest midpoint margin {margin.min():.2e}")
