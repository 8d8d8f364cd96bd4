"""Self-derived known-answer tests that pin the CPU oracle (the reference has no
golden vectors for this path; the reference-executed pins live in test_oracle_vs_reference.py, these are the
self-derived checks that came first): SE(3) exp/log, the cofactor
inverse, Huber, the bilinear gather's zero-blending quirk and finite-difference Jacobian checks."""
import numpy as np
import pytest

from oracle import oracle_np as ONP


def rand_pose(rng, ang=1.0, tr=0.5):
    w = rng.normal(size=3)
    w *= rng.uniform(0, ang) / np.linalg.norm(w)
    return w.astype(np.float32), rng.uniform(-tr, tr, 3).astype(np.float32)


def test_exp_log_roundtrip(oracle):
    rng = np.random.default_rng(0)
    for scale in (1e-5, 1e-3, 0.1, 1.0, 2.5, 3.1):
        for _ in range(20):
            w, u = rand_pose(rng, scale, 0.5)
            M = oracle.pose_to_matrix(w, u)
            R = M[:3, :3].astype(np.float64)
            assert np.abs(R @ R.T - np.eye(3)).max() < 5e-6
            assert abs(np.linalg.det(R) - 1) < 1e-5
            # fp32 (1 - cos t) / t^2 cancels badly for t just above the 1e-3 series threshold: the reference's
            # own formula is only good to ~1e-5 m there
            assert np.abs(M - ONP.se3_exp(w.astype(np.float64), u.astype(np.float64))).max() < 2e-5
            w2, u2 = oracle.matrix_to_pose(M)
            assert np.abs(w2 - w).max() < 2e-5 * max(1.0, scale) and np.abs(u2 - u).max() < 5e-5


def test_exp_branch_thresholds(oracle):
    """theta^2 just below/above 1e-8 and 1e-6 (LieDerivUtil.h:52,57,163,172): continuous across branches."""
    for th in (0.99e-4, 1.01e-4, 0.99e-3, 1.01e-3):
        w = np.array([th, 0, 0], np.float32)
        u = np.array([0.1, -0.2, 0.3], np.float32)
        M = oracle.pose_to_matrix(w, u)
        assert np.abs(M - ONP.se3_exp(w.astype(np.float64), u.astype(np.float64))).max() < 1e-5   # fp32 cancellation in (1-cos)/t^2 just above the series threshold


def test_log_near_pi_branch(oracle):
    """cos(angle) < -0.7071 branch of ln_rotation (LieDerivUtil.h:92-122)."""
    rng = np.random.default_rng(1)
    for _ in range(20):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        th = rng.uniform(2.6, 3.1)
        M = ONP.se3_exp(ax * th, np.zeros(3)).astype(np.float32)
        w, _ = oracle.matrix_to_pose(M)
        assert abs(np.linalg.norm(w) - th) < 2e-3 * 1.0
        # exp(log(R)) reproduces R; cos(angle) enters asin/acos with fp32 noise ~1e-7 -> angle noise ~1e-3 near pi
        assert np.abs(oracle.pose_to_matrix(w, np.zeros(3, np.float32))[:3, :3] - M[:3, :3]).max() < 3e-3


def test_inverse(oracle):
    rng = np.random.default_rng(2)
    for _ in range(50):
        w, u = rand_pose(rng)
        M = oracle.pose_to_matrix(w, u)
        Mi = oracle.mat4_inverse(M)
        assert np.abs(Mi.astype(np.float64) @ M.astype(np.float64) - np.eye(4)).max() < 2e-6
    G = rng.normal(size=(4, 4)).astype(np.float32)          # the inverse is generic, not the rigid shortcut
    assert np.abs(oracle.mat4_inverse(G).astype(np.float64) - np.linalg.inv(G.astype(np.float64))).max() < 1e-3 * np.abs(np.linalg.inv(G)).max()


def test_lie_update_is_left_composition(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        dW, dT = rand_pose(rng, 0.05, 0.01)
        cW, cT = rand_pose(rng, 1.0, 0.7)
        nW, nT = oracle.lie_update(dW, dT, cW, cT)
        want = ONP.se3_exp(dW.astype(np.float64), dT.astype(np.float64)) @ ONP.se3_exp(cW.astype(np.float64), cT.astype(np.float64))
        assert np.abs(oracle.pose_to_matrix(nW, nT) - want).max() < 3e-6


def test_huber(oracle):
    d = 0.005
    assert oracle.huber_weight(d * d, d) == 1.0                       # e <= delta^2 inclusive (SolverBundlingUtil.h:27)
    assert oracle.huber_weight(np.float32(d) * np.float32(d) * np.float32(1.0001), d) < 1.0
    for e in (1e-4, 1e-2, 4.0):
        assert abs(oracle.huber_weight(e, d) - d / np.sqrt(e)) < 1e-7


def test_bilinear_blends_zero_taps_and_skips_outside(oracle):
    img = np.zeros((4, 5, 4), np.float32)
    img[1, 1] = [1, 2, 3, 1]
    img[1, 2] = [0, 0, 0, 0]            # invalid tap = zeros, NOT -inf: blended in
    img[2, 1] = [3, 2, 1, 1]
    img[2, 2] = [1, 1, 1, 1]
    ok, v = oracle.bilinear4(1.25, 1.5, img)
    a, b = 0.25, 0.5
    want = (1 - b) * ((1 - a) * img[1, 1] + a * img[1, 2]) + b * ((1 - a) * img[2, 1] + a * img[2, 2])
    assert ok and np.allclose(v, want, atol=1e-6)
    # x in [-0.5, 0): floor = -1 -> left taps skipped through the unsigned compare, weights renormalised
    ok, v = oracle.bilinear4(-0.25, 1.0, img)
    assert ok and np.allclose(v, img[1, 0], atol=1e-6)
    # bottom row: y0+1 out of the image -> only the y0 row contributes
    ok, v = oracle.bilinear4(1.5, 3.0, img)
    assert ok and np.allclose(v, 0.5 * img[3, 1] + 0.5 * img[3, 2], atol=1e-6)
    # a -inf tap is skipped (never occurs in practice)
    img2 = img.copy(); img2[1, 1, 0] = -np.inf
    ok, v = oracle.bilinear4(1.25, 1.0, img2)
    assert ok and np.allclose(v, img2[1, 2], atol=1e-6)


def test_dense_jacobian_literal_equals_closed_form(oracle):
    """evalLie_derivI(Tj^-1, Ti, p) == -evalLie_derivJ(Ti^-1, Tj, p) and rows = [-n_w ; n_w x w]
    (LieDerivUtil.h:228-273, SolverBundlingEquationsLie.h:214-230)."""
    rng = np.random.default_rng(4)
    for _ in range(20):
        Ti = oracle.pose_to_matrix(*rand_pose(rng)); Tj = oracle.pose_to_matrix(*rand_pose(rng))
        Tii, Tji = oracle.mat4_inverse(Ti), oracle.mat4_inverse(Tj)
        p = rng.uniform(-0.3, 0.3, 3).astype(np.float32) + np.array([0, 0, 0.7], np.float32)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        JI = oracle.lie_deriv("I", Tji, Ti, p)
        JJ = oracle.lie_deriv("J", Tii, Tj, p)
        assert np.abs(JI + JJ).max() < 5e-6
        row_j = -(JJ.T @ n.astype(np.float32))
        w = (Tj.astype(np.float64) @ np.append(p, 1))[:3]
        n_w = Ti[:3, :3].astype(np.float64) @ n
        assert np.abs(row_j - np.concatenate([-n_w, np.cross(n_w, w)])).max() < 5e-6


def test_dense_row_is_derivative_of_residual():
    """Finite differences: res(delta) = (c_i - (Exp(d_i)T_i)^-1 Exp(d_j)T_j c_j) . n_i, left perturbation,
    per-frame order [trans, rot]."""
    rng = np.random.default_rng(5)
    Ti, Tj = ONP.se3_exp(rng.normal(size=3) * 0.4, rng.normal(size=3) * 0.3), ONP.se3_exp(rng.normal(size=3) * 0.4, rng.normal(size=3) * 0.3)
    cj, ci = rng.uniform(-0.2, 0.2, 3) + [0, 0, 0.7], rng.uniform(-0.2, 0.2, 3) + [0, 0, 0.7]
    n = rng.normal(size=3); n /= np.linalg.norm(n)

    def res(di, dj):
        A = ONP.se3_exp(di[3:], di[:3]) @ Ti
        B = ONP.se3_exp(dj[3:], dj[:3]) @ Tj
        q = (np.linalg.inv(A) @ B @ np.append(cj, 1))[:3]
        return (ci - q) @ n            # n_i, c_i held fixed (they are looked up, not differentiated)

    w = (Tj @ np.append(cj, 1))[:3]
    n_w = Ti[:3, :3] @ n
    row_j = np.concatenate([-n_w, np.cross(n_w, w)])
    h = 1e-6
    for k in range(6):
        e = np.zeros(6); e[k] = h
        fd_j = (res(np.zeros(6), e) - res(np.zeros(6), -e)) / (2 * h)
        fd_i = (res(e, np.zeros(6)) - res(-e, np.zeros(6))) / (2 * h)
        assert abs(fd_j - row_j[k]) < 1e-6 and abs(fd_i + row_j[k]) < 1e-6


def test_sparse_jacobian_is_derivative_of_residual():
    """r = T_i p_i - T_j p_j; d r / d(rot_i) columns = evalLie_dAlpha/dBeta/dGamma(w_i) (LieDerivUtil.h:215-226)."""
    rng = np.random.default_rng(6)
    Ti = ONP.se3_exp(rng.normal(size=3) * 0.4, rng.normal(size=3) * 0.3)
    p = rng.uniform(-0.1, 0.1, 3) + [0, 0, 0.7]
    w = (Ti @ np.append(p, 1))[:3]
    cols = np.stack([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], 1)
    h = 1e-6
    for k in range(3):
        e = np.zeros(3); e[k] = h
        fd = ((ONP.se3_exp(e, np.zeros(3)) @ Ti @ np.append(p, 1))[:3] - (ONP.se3_exp(-e, np.zeros(3)) @ Ti @ np.append(p, 1))[:3]) / (2 * h)
        assert np.abs(fd - cols[:, k]).max() < 1e-7
        fdt = ((ONP.se3_exp(np.zeros(3), e) @ Ti @ np.append(p, 1))[:3] - (ONP.se3_exp(np.zeros(3), -e) @ Ti @ np.append(p, 1))[:3]) / (2 * h)
        assert np.abs(fdt - np.eye(3)[:, k]).max() < 1e-7
