"""fp64 numpy restatement of the same solver, written independently of btba_oracle.c.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  Purpose: sanity-check the literal fp32 C
oracle with a second derivation that (a) uses the closed-form dense Jacobian rows
row_j = [-n_w ; n_w x w], row_i = -row_j instead of the
reference's 3x12 * 12x6 product (LieDerivUtil.h:228-273), and (b) forms the sparse normal
matrix explicitly from per-pair moment sums instead of applying J and J^T matrix-free
(SolverBundlingEquationsLie.h:140-211).  Both are exactly the algebra the HIP kernels use,
so this file is also the executable specification of the kernels' math.
"""
from __future__ import annotations

import numpy as np

EPS = 1e-6  # FLOAT_EPSILON, SolverUtil.h:10


def skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def se3_exp(w, u):
    th = np.linalg.norm(w)
    K = skew(w)
    if th < 1e-10:
        R, V = np.eye(3) + K, np.eye(3) + 0.5 * K
    else:
        A, B, C = np.sin(th) / th, (1 - np.cos(th)) / th**2, (th - np.sin(th)) / th**3
        R, V = np.eye(3) + A * K + B * K @ K, np.eye(3) + B * K + C * K @ K
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, V @ u
    return T


def so3_log(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    v = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(v)
    if s < 1e-12:
        return v
    return v * (np.arctan2(s, c) / s)


def se3_log(T):
    w = so3_log(T[:3, :3])
    th = np.linalg.norm(w)
    K = skew(w)
    if th < 1e-10:
        V = np.eye(3) + 0.5 * K
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * K + (th - np.sin(th)) / th**3 * K @ K
    return w, np.linalg.solve(V, T[:3, 3])


def huber_w(e, delta):
    e = np.asarray(e, np.float64)
    return np.where(e <= delta * delta, 1.0, delta / np.sqrt(np.maximum(e, 1e-300)))


def bilinear4(img, u, v):
    """ICPUtil.h:83-110 vectorised: img [H,W,4], u,v [n].  Invalid (zero) taps blend in."""
    H, W = img.shape[:2]
    x0, y0 = np.floor(u).astype(np.int64), np.floor(v).astype(np.int64)
    al, be = u - x0, v - y0

    def tap(x, y):
        ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
        val = img[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)].astype(np.float64)
        return ok, val

    o00, v00 = tap(x0, y0); o10, v10 = tap(x0 + 1, y0); o01, v01 = tap(x0, y0 + 1); o11, v11 = tap(x0 + 1, y0 + 1)
    w0 = o00 * (1 - al) + o10 * al
    s0 = (o00 * (1 - al))[:, None] * v00 + (o10 * al)[:, None] * v10
    w1 = o01 * (1 - al) + o11 * al
    s1 = (o01 * (1 - al))[:, None] * v01 + (o11 * al)[:, None] * v11
    with np.errstate(divide="ignore", invalid="ignore"):
        p0, p1 = s0 / w0[:, None], s1 / w1[:, None]
        u0, u1 = w0 > 0, w1 > 0
        ww = u0 * (1 - be) + u1 * be
        ss = np.where(u0[:, None], (1 - be)[:, None] * p0, 0) + np.where(u1[:, None], be[:, None] * p1, 0)
        out = ss / ww[:, None]
    return ww > 0, out


def dense_pair_sums(campos, normals, intr, Ti, Tj, Tinv_i, prm):
    """S = sum w a a^T (6x6), g = sum w a res (6), count; a = row_j in [trans, rot] order."""
    fx, fy, cx, cy = [float(v) for v in intr]
    cs4 = campos[1].reshape(-1, 4).astype(np.float64)      # source
    ns4 = normals[1].reshape(-1, 4).astype(np.float64)
    H, W = campos[0].shape[:2]
    ok = (cs4[:, 2] > prm["depth_min"]) & (cs4[:, 2] < prm["depth_max"])
    Tij = Tinv_i @ Tj
    cs = cs4[:, :3]
    q = cs @ Tij[:3, :3].T + Tij[:3, 3]
    nq = ns4[:, :3] @ Tij[:3, :3].T
    with np.errstate(divide="ignore", invalid="ignore"):
        u = q[:, 0] * fx / q[:, 2] + cx
        v = q[:, 1] * fy / q[:, 2] + cy
    ru, rv = np.sign(u) * np.floor(np.abs(u) + 0.5), np.sign(v) * np.floor(np.abs(v) + 0.5)   # roundf
    ok &= np.isfinite(u) & np.isfinite(v) & (ru >= 0) & (rv >= 0) & (ru < W) & (rv < H)
    u, v = np.where(ok, u, 0.0), np.where(ok, v, 0.0)
    vc, ci = bilinear4(campos[0], u, v)
    vn, ni = bilinear4(normals[0], u, v)
    ok &= vc & vn & (ci[:, 2] > prm["depth_min"]) & (ci[:, 2] < prm["depth_max"])
    dist = np.linalg.norm(q - ci[:, :3], axis=1)
    dn = (nq * ni[:, :3]).sum(1)
    ok &= (dn >= prm["dense_normal_thresh"]) & (dist <= prm["dense_dist_thresh"])
    res = ((ci[:, :3] - q) * ni[:, :3]).sum(1)
    wgt = prm["weight_dense_depth"] * huber_w(res * res, prm["robust_delta"])
    w_world = cs @ Tj[:3, :3].T + Tj[:3, 3]
    n_w = ni[:, :3] @ Ti[:3, :3].T
    a = np.concatenate([-n_w, np.cross(n_w, w_world)], 1)[ok]
    wgt, res = wgt[ok], res[ok]
    S = (a * wgt[:, None]).T @ a
    g = (a * (wgt * res)[:, None]).sum(0)
    return S, g, int(ok.sum())


def solve(campos, normals, intr, corr, poses, pairs=None, **kw):
    prm = dict(n_gn_iters=7, n_pcg_iters=5, robust_delta=0.005, dense_dist_thresh=0.02,
               dense_normal_thresh=float(np.float32(np.cos(np.pi / 4))), depth_min=0.1, depth_max=9999.0,
               weight_sparse=1.0, weight_dense_depth=1.0)
    prm.update(kw)
    N = poses.shape[0]
    if pairs is None:
        pairs = [(i, j) for i in range(N) for j in range(i + 1, N)]
    x = [se3_log(np.asarray(poses[k], np.float64)) for k in range(N)]
    ci, cj = corr["imgIdx_i"].astype(np.int64), corr["imgIdx_j"].astype(np.int64)
    valid = corr["imgIdx_i"] != 0xFFFFFFFF
    pi, pj = corr["pos_i"].astype(np.float64), corr["pos_j"].astype(np.float64)
    out = dict(T_after=[], x_after=[], dense_count=[], pcg_scalars=[], A=[], b=[])
    dim = 6 * N
    for it in range(prm["n_gn_iters"]):
        T = np.stack([se3_exp(*x[k]) for k in range(N)])
        Tinv = np.linalg.inv(T)
        A = np.zeros((dim, dim)); b = np.zeros(dim); Mdiag = np.zeros(dim)
        cnts = []
        ws, wd = prm["weight_sparse"], prm["weight_dense_depth"]
        if wd > 0:
            for (i, j) in pairs:
                S, g, cnt = dense_pair_sums(campos[[i, j]], normals[[i, j]], intr, T[i], T[j], Tinv[i], prm)
                cnts.append(cnt)
                if j > 0:
                    A[6 * j:6 * j + 6, 6 * j:6 * j + 6] += S; b[6 * j:6 * j + 6] -= g
                if i > 0:
                    A[6 * i:6 * i + 6, 6 * i:6 * i + 6] += S; b[6 * i:6 * i + 6] += g
                if i > 0 and j > 0 and i < j:       # i>j: cross block erased by FlipJtJ (appendix A.6)
                    A[6 * j:6 * j + 6, 6 * i:6 * i + 6] -= S; A[6 * i:6 * i + 6, 6 * j:6 * j + 6] -= S.T
        # sparse: per correspondence J_i = [I | -[w_i]x] in [trans, rot] order
        if ws > 0 and valid.any():
            wi = np.einsum("nab,nb->na", T[ci[valid], :3, :3], pi[valid]) + T[ci[valid], :3, 3]
            wj = np.einsum("nab,nb->na", T[cj[valid], :3, :3], pj[valid]) + T[cj[valid], :3, 3]
            r = wi - wj
            rho = huber_w((r * r).sum(1), prm["robust_delta"])
            for k in range(1, N):
                for (sel, w, sign) in ((ci[valid] == k, wi, 1.0), (cj[valid] == k, wj, -1.0)):
                    if not sel.any():
                        continue
                    wk, rk, rh = w[sel], r[sel], rho[sel]
                    b[6 * k:6 * k + 3] += -ws * sign * (rh[:, None] * rk).sum(0)
                    b[6 * k + 3:6 * k + 6] += -ws * sign * (rh[:, None] * np.cross(wk, rk)).sum(0)
                    Mdiag[6 * k:6 * k + 3] += rh.sum()
                    Mdiag[6 * k + 3:6 * k + 6] += (rh[:, None] * np.stack([wk[:, 1]**2 + wk[:, 2]**2, wk[:, 0]**2 + wk[:, 2]**2, wk[:, 0]**2 + wk[:, 1]**2], 1)).sum(0)
            # lhs = ws * J^T J (unweighted by rho)
            def Jblk(w):
                n = w.shape[0]
                J = np.zeros((n, 3, 6)); J[:, :, :3] = np.eye(3)
                J[:, 0, 4], J[:, 0, 5] = w[:, 2], -w[:, 1]
                J[:, 1, 3], J[:, 1, 5] = -w[:, 2], w[:, 0]
                J[:, 2, 3], J[:, 2, 4] = w[:, 1], -w[:, 0]
                return J
            Ji, Jj = Jblk(wi), Jblk(wj)
            vi, vj = ci[valid], cj[valid]
            for (a_, b_) in set(zip(vi.tolist(), vj.tolist())):
                sel = (vi == a_) & (vj == b_)
                if a_ > 0:
                    A[6 * a_:6 * a_ + 6, 6 * a_:6 * a_ + 6] += ws * np.einsum("nka,nkb->ab", Ji[sel], Ji[sel])
                if b_ > 0:
                    A[6 * b_:6 * b_ + 6, 6 * b_:6 * b_ + 6] += ws * np.einsum("nka,nkb->ab", Jj[sel], Jj[sel])
                if a_ > 0 and b_ > 0:
                    X = ws * np.einsum("nka,nkb->ab", Ji[sel], Jj[sel])
                    A[6 * a_:6 * a_ + 6, 6 * b_:6 * b_ + 6] -= X; A[6 * b_:6 * b_ + 6, 6 * a_:6 * a_ + 6] -= X.T
        M = np.where(Mdiag > EPS, 1.0 / np.where(Mdiag > EPS, Mdiag, 1.0), 1.0)
        act = np.arange(6, dim)
        Aa, ba, Ma = A[np.ix_(act, act)], b[act], M[act]
        d = np.zeros_like(ba); rr = ba.copy(); z = Ma * rr; p = z.copy(); rz = rr @ z
        sc = []
        for li in range(prm["n_pcg_iters"]):
            Ap = Aa @ p
            pAp = p @ Ap
            alpha = rz / pAp if pAp > EPS else 0.0
            d += alpha * p; rr -= alpha * Ap; z = Ma * rr; rzn = z @ rr
            beta = rzn / rz if rz > EPS else 0.0
            sc.append((pAp, alpha, rzn, beta))
            rz = rzn; p = z + beta * p
        for k in range(1, N):
            dk = d[6 * (k - 1):6 * k]
            x[k] = se3_log(se3_exp(dk[3:], dk[:3]) @ se3_exp(*x[k]))
        out["T_after"].append(np.stack([se3_exp(*x[k]) for k in range(N)]))
        out["x_after"].append(np.stack([np.concatenate(x[k]) for k in range(N)]))
        out["dense_count"].append(cnts); out["pcg_scalars"].append(sc); out["A"].append(A); out["b"].append(b)
    out["poses"] = out["T_after"][-1]
    return out
