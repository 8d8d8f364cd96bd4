"""ctypes binding of oracle/_ref/libbtba_ref.so: the REFERENCE'S OWN device functions compiled for the CPU
(oracle/ref_driver.cpp + oracle/ref_shim/, built by `make -C oracle ref` where /root/reference exists).
Test infrastructure: used only by tests/ to pin oracle/btba_oracle.c against the reference itself."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libbtba_ref.so")
REFERENCE = os.environ.get("BTBA_REFERENCE", "/root/reference")
_lib = None


def available() -> bool:
    """True when the library exists or can be built (the reference checkout is present)."""
    return os.path.exists(SO) or os.path.isdir(os.path.join(REFERENCE, "src", "cuda", "Solver"))


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, "ref_driver.cpp"), os.path.join(_HERE, "ref_shim", "cuda_runtime.h")]
    stale = not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in srcs)
    if (force or stale) and os.path.isdir(os.path.join(REFERENCE, "src", "cuda", "Solver")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref", "REFERENCE=" + REFERENCE] + (["-B"] if force else []))
    return SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(*vals):
    return [np.ascontiguousarray(v, np.float32) for v in vals]


def pose_to_matrix(rot, trans):
    r, t = _f(rot, trans); M = np.zeros(16, np.float32)
    lib().ref_pose_to_matrix(_p(r), _p(t), _p(M))
    return M.reshape(4, 4)


def matrix_to_pose(M):
    (M,) = _f(np.reshape(M, 16)); r, t = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().ref_matrix_to_pose(_p(M), _p(r), _p(t))
    return r, t


def mat4_inverse(M):
    (M,) = _f(np.reshape(M, 16)); o = np.zeros(16, np.float32)
    lib().ref_mat4_inverse(_p(M), _p(o))
    return o.reshape(4, 4)


def lie_update(dW, dT, cW, cT):
    a = _f(dW, dT, cW, cT); nW, nT = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().ref_lie_update(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(nW), _p(nT))
    return nW, nT


def lie_deriv(which, A, D, p):
    A, D, p = _f(np.reshape(A, 16), np.reshape(D, 16), p); jac = np.zeros(18, np.float32)
    lib().ref_lie_deriv(1 if which == "J" else 0, _p(A), _p(D), _p(p), _p(jac))
    return jac.reshape(3, 6)


def bilinear4(x, y, img):
    (img,) = _f(img); H, W = img.shape[:2]; out = np.zeros(4, np.float32)
    f = lib().ref_bilinear4; f.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    f(float(x), float(y), _p(img), W, H, _p(out))
    return out


def huber(e, delta):
    rho = np.zeros(3, np.float32)
    f = lib().ref_huber; f.argtypes = [C.c_float, C.c_float, C.c_void_p]
    f(float(e), float(delta), _p(rho))
    return rho


def sparse_rhs(corr, T, robust_delta=0.005, weight_sparse=1.0):
    T = np.ascontiguousarray(T, np.float32); N = T.shape[0]
    corr = np.ascontiguousarray(corr)
    rhs, prec = np.zeros((N, 6), np.float32), np.zeros((N, 6), np.float32)
    f = lib().ref_sparse_rhs; f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    f(N, _p(corr), len(corr), _p(T), robust_delta, weight_sparse, _p(rhs), _p(prec))
    return rhs, prec


def sparse_apply(corr, T, p, weight_sparse=1.0):
    T = np.ascontiguousarray(T, np.float32); N = T.shape[0]
    corr = np.ascontiguousarray(corr); p = np.ascontiguousarray(p, np.float32).reshape(N, 6)
    out = np.zeros((N, 6), np.float32)
    f = lib().ref_sparse_apply; f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    f(N, _p(corr), len(corr), _p(T), weight_sparse, _p(p), _p(out))
    return out


def dense_system(campos, normals, intr, T, Tinv, pairs, dist_thresh=0.02, normal_thresh=float(np.cos(np.pi / 4)), depth_min=0.1, depth_max=9999.0,
                 robust_delta=0.005, weight_dense=1.0):
    campos, normals, T, Tinv = _f(campos, normals, T, Tinv)
    N, Hd, Wd = campos.shape[:3]
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    (intr,) = _f(intr)
    JtJ, Jtr, cnt = np.zeros((6 * N, 6 * N), np.float32), np.zeros(6 * N, np.float32), np.zeros(len(pairs), np.int32)
    f = lib().ref_dense_system
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_float] * 6 + [C.c_void_p] * 3
    f(N, Wd, Hd, _p(intr), _p(campos), _p(normals), _p(T), _p(Tinv), _p(pairs), len(pairs), dist_thresh, normal_thresh, depth_min, depth_max, robust_delta, weight_dense,
      _p(JtJ), _p(Jtr), _p(cnt))
    return JtJ, Jtr, cnt


# ---- the reference's procrustesKernel / evalPoseKernel (oracle/_ref/libbtba_ref_ransac.so) -------------------
SO_RANSAC = os.path.join(_HERE, "_ref", "libbtba_ref_ransac.so")
_lib_r = None


def lib_ransac() -> C.CDLL:
    global _lib_r
    if _lib_r is None:
        if not os.path.exists(SO_RANSAC):
            build(force=True)
        _lib_r = C.CDLL(SO_RANSAC)
    return _lib_r


def _pts4(p):
    p = np.asarray(p, np.float32)
    if p.shape[1] == 3:
        p = np.concatenate([p, np.ones((p.shape[0], 1), np.float32)], 1)
    return np.ascontiguousarray(p, np.float32)


def procrustes(src, dst):
    """procrustesKernel (cuda_ransac.cu:999-1102, with the reference's approximate 3x3 SVD): (ok, pose [4,4])."""
    s, d = _pts4(src), _pts4(dst); pose = np.zeros(16, np.float32)
    f = lib_ransac().ref_procrustes; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    ok = f(_p(s), _p(d), s.shape[0], _p(pose))
    return bool(ok), pose.reshape(4, 4)


def eval_pose(ptsA, ptsB, pose, dist_thres):
    """evalPoseKernel (cuda_ransac.cu:978-997): ascending ids of the points with |ptB - pose ptA| <= dist_thres."""
    a, b = _pts4(ptsA), _pts4(ptsB); ids = np.zeros(max(len(a), 1), np.int32)
    (P,) = _f(np.reshape(pose, 16))
    f = lib_ransac().ref_eval_pose; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p]
    n = f(_p(a), _p(b), len(a), _p(P), dist_thres, _p(ids))
    return ids[:n].copy()


def ransac_multi_pair(ptsA, ptsB, n_trials, dist_thres):
    """ransacMultiPairGPU (cuda_ransac.cu:1228-1323) end to end -- ransacEstimateModelKernel (cuRAND triples via oracle/xorwow.h,
    procrustesKernel), ransacEvalModelKernel, findBestTrial, the host-side inlier gather -- on the sequential emulator.
    ptsA[p], ptsB[p]: [n_p, 3|4] points of pair p.  Returns one ascending int32 inlier-id array per pair.  Among trials tied for
    the most inliers the emulation keeps the LAST one (see oracle/ref_ransac_wrap.h)."""
    A, B = [_pts4(a) for a in ptsA], [_pts4(b) for b in ptsB]
    n_pts = np.array([len(a) for a in A], np.int32)
    a_all, b_all = np.ascontiguousarray(np.concatenate(A)), np.ascontiguousarray(np.concatenate(B))
    ids = np.zeros(max(int(n_pts.sum()), 1), np.int32); n_in = np.zeros(len(A), np.int32)
    f = lib_ransac().ref_ransac_multi_pair
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    f(_p(a_all), _p(b_all), _p(n_pts), len(A), int(n_trials), float(dist_thres), _p(ids), _p(n_in))
    out, o = [], 0
    for p in range(len(A)):
        out.append(ids[o:o + n_in[p]].copy()); o += int(n_pts[p])
    return out


# ---- the reference's WHOLE solver, emulated sequentially (oracle/_ref/libbtba_ref_solver.so) -----------------
SO_SOLVER = os.path.join(_HERE, "_ref", "libbtba_ref_solver.so")
SO_SOLVER_FM = os.path.join(_HERE, "_ref", "libbtba_ref_solver_fm.so")      # the same sources under a model of the reference's own build flags (-use_fast_math)
_lib_s = {}


def lib_solver(fastmath: bool = False) -> C.CDLL:
    so = SO_SOLVER_FM if fastmath else SO_SOLVER
    if fastmath not in _lib_s:
        if not os.path.exists(so):
            build(force=True)
        L = C.CDLL(so)
        L.ref_set_order.argtypes = [C.c_int, C.c_ulonglong]
        assert L.ref_build_flags() == (1 if fastmath else 0)
        if fastmath:
            L.ref_set_fastmath_seed.argtypes = [C.c_ulonglong]
        _lib_s[fastmath] = L
    return _lib_s[fastmath]


def _order_args(order):
    """'forward' | 'reverse' | 'shuffle:<seed>' | None (= follow the environment variable BTBA_REF_ORDER) -> (mode, seed) of ref_set_order."""
    if order is None:
        return -1, 0
    if order == "forward":
        return 0, 0
    if order == "reverse":
        return 1, 0
    if order.startswith("shuffle"):
        return 2, int(order.split(":")[1]) if ":" in order else 1
    raise ValueError(order)


def solve(campos, normals, intr, corr, poses, n_gn=7, n_pcg=5, weight_sparse=1.0, weight_dense=1.0, robust_delta=0.005,
          dist_thresh=0.02, normal_thresh=float(np.cos(np.pi / 4)), depth_min=0.1, depth_max=9999.0, addr_rank=None,
          weights_sparse=None, weights_dense=None, order="forward", fastmath=False, fastmath_seed=1, want_iterates=False):
    """solveBundlingStub (SolverBundling.cu:931-1003) and everything under it, run by the reference's own code.
    Returns (poses [N,4,4] after n_gn Gauss-Newton iterations, x [N,6] = (rot, trans)); with want_iterates a third value,
    T_iter [n_gn, N, 4, 4] = the reference's poseToMatrix of its unknowns after every iteration (recorded at the top of the next one).
    addr_rank: permutation of 0..N-1 ordering the frames' d_num_valid_points ADDRESSES, which is what orients the dense pairs in
    the reference (FindImageImageCorr_Kernel keeps (target i, source j) iff address_i > address_j): None = descending in frame
    order (target = lower index), np.arange(N) = ascending (target = higher index, cross blocks erased by FlipJtJ).
    weights_sparse / weights_dense: per-iteration weights [n_gn] (input.weightsSparse / weightsDenseDepth, SolverBundling.cu:948-949) or None.
    order: the order in which the launch emulator runs the (block, thread) cells of every launch -- each a legal execution of kernels that
    meet only through float atomics (ref_shim/cuda_runtime.h); fastmath: the library built under the model of -use_fast_math (oracle/Makefile)."""
    campos, normals = _f(campos, normals)
    N, Hd, Wd = campos.shape[:3]
    (intr,) = _f(intr)
    corr = np.ascontiguousarray(corr)
    P = np.ascontiguousarray(poses, np.float32).reshape(N, 16).copy()
    x = np.zeros((N, 6), np.float32)
    L = lib_solver(fastmath)
    L.ref_set_order(*_order_args(order))
    if fastmath:
        L.ref_set_fastmath_seed(int(fastmath_seed))
    f = L.ref_solve4
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 7 + [C.c_void_p] * 5
    wsi = np.ascontiguousarray(weights_sparse, np.float32) if weights_sparse is not None else None
    wdi = np.ascontiguousarray(weights_dense, np.float32) if weights_dense is not None else None
    assert (wsi is None or len(wsi) == n_gn) and (wdi is None or len(wdi) == n_gn)
    rank = None
    if addr_rank is not None:
        rank = np.ascontiguousarray(addr_rank, np.int32)
        if sorted(rank.tolist()) != list(range(N)):
            raise ValueError("addr_rank must be a permutation of 0..N-1")
    T_iter = np.zeros((n_gn, N, 4, 4), np.float32) if want_iterates else None
    f(N, Wd, Hd, _p(intr), _p(campos), _p(normals), _p(corr), len(corr), _p(P), n_gn, n_pcg, weight_sparse, weight_dense, robust_delta,
      dist_thresh, normal_thresh, depth_min, depth_max, _p(x), _p(rank) if rank is not None else None,
      _p(wsi) if wsi is not None else None, _p(wdi) if wdi is not None else None, _p(T_iter) if want_iterates else None)
    if want_iterates:
        return P.reshape(N, 4, 4), x, T_iter
    return P.reshape(N, 4, 4), x


# (execution order, fast-math model) of the runs a self-spread measurement compares; the first entry is the run rounds 1-5 used
VARIANTS = [("forward", False), ("reverse", False), ("shuffle:1", False), ("shuffle:2", False),
            ("forward", True), ("reverse", True), ("shuffle:1", True), ("shuffle:2", True)]


def self_spread(campos, normals, intr, corr, poses, pose_error, variants=None, **kw):
    """THE REFERENCE AGAINST ITSELF: the same inputs through the reference's own solver under several legal execution orders of its float
    atomics and, for the fast-math entries, under the model of its own -use_fast_math build.  Returns (spread [n_gn] = per iterate the largest
    pose difference (rad | m) between any variant and the forward / IEEE run, the iterates of every run [n_variants, n_gn, N, 4, 4])."""
    variants = VARIANTS if variants is None else variants
    N = np.asarray(poses).reshape(-1, 4, 4).shape[0]
    runs = [solve(campos, normals, intr, corr, poses, order=o, fastmath=fm, want_iterates=True, **kw)[2] for (o, fm) in variants]
    base = runs[0]
    G = base.shape[0]
    spread = np.zeros(G)
    for r in runs[1:]:
        for it in range(G):
            spread[it] = max(spread[it], max(max(pose_error(r[it, k], base[it, k])) for k in range(N)))
    return spread, np.stack(runs)


# ---- the reference's image kernels (oracle/_ref/libbtba_ref_image.so) ----------------------------------------
SO_IMAGE = os.path.join(_HERE, "_ref", "libbtba_ref_image.so")
_lib_i = None


def lib_image() -> C.CDLL:
    global _lib_i
    if _lib_i is None:
        if not os.path.exists(SO_IMAGE):
            build(force=True)
        _lib_i = C.CDLL(SO_IMAGE)
    return _lib_i


def store_frame(depth, normals, Kinv4, downscale=4.0):
    """CUDACache::storeFrame (CUDACache.cpp:76-88) by the reference's kernels: (campos, normals, depth, n_valid) at cache size."""
    depth, normals, Kinv4 = _f(depth, normals, np.reshape(Kinv4, 16))
    H, W = depth.shape
    Wd, Hd = int(W / downscale), int(H / downscale)
    cam, nrm, dd = np.zeros((Hd, Wd, 4), np.float32), np.zeros((Hd, Wd, 4), np.float32), np.zeros((Hd, Wd), np.float32)
    nv = C.c_int(0)
    f = lib_image().ref_store_frame
    f.argtypes = [C.c_int] * 4 + [C.c_void_p] * 6 + [C.POINTER(C.c_int)]
    f(W, H, Wd, Hd, _p(Kinv4), _p(depth), _p(normals), _p(cam), _p(nrm), _p(dd), C.byref(nv))
    return cam, nrm, dd, int(nv.value)


def process_depth(depth, erode_radius=1, erode_diff=0.001, erode_ratio=0.8, bf_radius=2, sigma_d=2.0, sigma_r=100000.0):
    (depth,) = _f(depth); H, W = depth.shape; out = np.zeros_like(depth)
    f = lib_image().ref_process_depth
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
    f(W, H, _p(depth), _p(out), int(erode_radius), erode_diff, erode_ratio, int(bf_radius), sigma_d, sigma_r)
    return out


def depth_to_normals(depth, Kinv4):
    depth, Kinv4 = _f(depth, np.reshape(Kinv4, 16)); H, W = depth.shape
    nrm, xyz = np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.float32)
    f = lib_image().ref_depth_to_normals
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f(W, H, _p(Kinv4), _p(depth), _p(nrm), _p(xyz))
    return nrm, xyz


def pairs_from_addr_rank(addr_rank):
    """The ordered (target, source) list FindImageImageCorr_Kernel emits for a given address order: (i, j) iff rank_i > rank_j,
    listed in canonical pair order (the reference's own list order is whatever its atomicAdd hands out)."""
    r = list(addr_rank)
    N = len(r)
    return np.array([(i, j) if r[i] > r[j] else (j, i) for i in range(N) for j in range(i + 1, N)], np.int32)
