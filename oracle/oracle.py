"""ctypes binding of the CPU oracle (oracle/btba_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from bundletrack_amd/.  Parity: pinned against the
reference's own solver compiled and executed on the CPU (oracle/reference.py, oracle/_ref,
tests/test_oracle_vs_reference.py); the reference has no golden vectors for this path
(SURVEY.md section 8c); see the header of btba_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbtba_oracle.so")

ENTRYJ_DTYPE = np.dtype(
    [("imgIdx_i", "<u4"), ("imgIdx_j", "<u4"), ("pos_i", "<f4", (3,)), ("pos_j", "<f4", (3,))]
)
assert ENTRYJ_DTYPE.itemsize == 32


class OrcParams(C.Structure):
    _fields_ = [
        ("n_gn_iters", C.c_int32),
        ("n_pcg_iters", C.c_int32),
        ("robust_delta", C.c_float),
        ("dense_dist_thresh", C.c_float),
        ("dense_normal_thresh", C.c_float),
        ("depth_min", C.c_float),
        ("depth_max", C.c_float),
        ("weight_sparse", C.c_float),
        ("weight_dense_depth", C.c_float),
        ("accum_mode", C.c_int32),
        ("n_threads", C.c_int32),
    ]


class OrcTrace(C.Structure):
    _fields_ = [
        ("x_after", C.c_void_p),
        ("T_after", C.c_void_p),
        ("dense_JtJ", C.c_void_p),
        ("dense_Jtr", C.c_void_p),
        ("rhs", C.c_void_p),
        ("precond", C.c_void_p),
        ("pcg_scalars", C.c_void_p),
        ("dense_count", C.c_void_p),
        ("delta", C.c_void_p),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("btba_oracle.c", "btba_oracle_ransac.c", "btba_oracle_keyframes.c", "xorwow.h", "Makefile"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_solve.restype = C.c_int
        _lib.orc_build_cache.restype = C.c_int
        _lib.orc_huber_weight.restype = C.c_float
        _lib.orc_huber_weight.argtypes = [C.c_float, C.c_float]
        _lib.orc_bilinear4.restype = C.c_int
        _lib.orc_bilinear4.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def default_params(**kw) -> OrcParams:
    p = OrcParams()
    lib().orc_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def pose_to_matrix(rot, trans) -> np.ndarray:
    rot = np.ascontiguousarray(rot, np.float32)
    trans = np.ascontiguousarray(trans, np.float32)
    M = np.zeros(16, np.float32)
    lib().orc_pose_to_matrix(_p(rot), _p(trans), _p(M))
    return M.reshape(4, 4)


def matrix_to_pose(M):
    M = np.ascontiguousarray(M, np.float32).reshape(16)
    r = np.zeros(3, np.float32)
    t = np.zeros(3, np.float32)
    lib().orc_matrix_to_pose(_p(M), _p(r), _p(t))
    return r, t


def mat4_inverse(M) -> np.ndarray:
    M = np.ascontiguousarray(M, np.float32).reshape(16)
    o = np.zeros(16, np.float32)
    lib().orc_mat4_inverse(_p(M), _p(o))
    return o.reshape(4, 4)


def lie_update(dW, dT, cW, cT):
    a = [np.ascontiguousarray(v, np.float32) for v in (dW, dT, cW, cT)]
    nW = np.zeros(3, np.float32)
    nT = np.zeros(3, np.float32)
    lib().orc_lie_update(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(nW), _p(nT))
    return nW, nT


def lie_deriv(which: str, A, D, p) -> np.ndarray:
    A = np.ascontiguousarray(A, np.float32).reshape(16)
    D = np.ascontiguousarray(D, np.float32).reshape(16)
    p = np.ascontiguousarray(p, np.float32)
    jac = np.zeros(18, np.float32)
    fn = lib().orc_lie_deriv_I if which == "I" else lib().orc_lie_deriv_J
    fn(_p(A), _p(D), _p(p), _p(jac))
    return jac.reshape(3, 6)


def bilinear4(x, y, img) -> tuple[int, np.ndarray]:
    img = np.ascontiguousarray(img, np.float32)
    H, W = img.shape[:2]
    out = np.zeros(4, np.float32)
    ok = lib().orc_bilinear4(float(x), float(y), _p(img), W, H, _p(out))
    return ok, out


def huber_weight(e, delta) -> float:
    return float(lib().orc_huber_weight(float(e), float(delta)))


def build_cache(depth: np.ndarray, normals: np.ndarray, K: np.ndarray, downscale: float = 4.0):
    """CUDACache::storeFrame for one frame.  depth [H,W] f32, normals [H,W,4] f32, K [3,3].
    Returns dict(campos [Hd,Wd,4], normals [Hd,Wd,4], depth [Hd,Wd], n_valid, intr (fx,fy,cx,cy))."""
    depth = np.ascontiguousarray(depth, np.float32)
    normals = np.ascontiguousarray(normals, np.float32)
    H, W = depth.shape
    Wd, Hd = int(W / downscale), int(H / downscale)      # LossGPU.cu:56-57
    Kf = np.ascontiguousarray(K, np.float32).reshape(9)
    campos = np.zeros((Hd, Wd, 4), np.float32)
    nrm = np.zeros((Hd, Wd, 4), np.float32)
    dd = np.zeros((Hd, Wd), np.float32)
    nv = C.c_int32(0)
    intr = np.zeros(4, np.float32)
    rc = lib().orc_build_cache(H, W, Hd, Wd, _p(Kf), _p(depth), _p(normals), _p(campos), _p(nrm), _p(dd), C.byref(nv), _p(intr))
    if rc != 0:
        raise ValueError("orc_build_cache failed")
    return dict(campos=campos, normals=nrm, depth=dd, n_valid=int(nv.value), intr=intr)


@dataclass
class Trace:
    x_after: np.ndarray
    T_after: np.ndarray
    dense_JtJ: np.ndarray
    dense_Jtr: np.ndarray
    rhs: np.ndarray
    precond: np.ndarray
    pcg_scalars: np.ndarray
    dense_count: np.ndarray
    delta: np.ndarray
    poses: np.ndarray = field(default=None)


def target_lower_pairs(n_frames: int) -> np.ndarray:
    """TARGET_LOWER dense pair policy: every i<j once, target = i."""
    return np.array([(i, j) for i in range(n_frames) for j in range(i + 1, n_frames)], np.int32).reshape(-1, 2)


def solve(campos, normals, intr, corr, poses, pairs=None, params: OrcParams | None = None, want_trace=True) -> Trace:
    """campos/normals [N,Hd,Wd,4]; intr (fx,fy,cx,cy) downscaled; corr ENTRYJ_DTYPE[C];
    poses [N,4,4] row-major camera->model.  Returns Trace with .poses = optimised poses."""
    campos = np.ascontiguousarray(campos, np.float32)
    normals = np.ascontiguousarray(normals, np.float32)
    N, Hd, Wd = campos.shape[:3]
    prm = params or default_params()
    corr = np.ascontiguousarray(corr, ENTRYJ_DTYPE)
    if pairs is None:
        pairs = target_lower_pairs(N)
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    P = pairs.shape[0]
    poses_io = np.ascontiguousarray(poses, np.float32).reshape(N, 16).copy()
    intr = np.ascontiguousarray(intr, np.float32)
    G, L, dim = prm.n_gn_iters, prm.n_pcg_iters, 6 * N
    t = Trace(
        x_after=np.zeros((G, N, 6), np.float32), T_after=np.zeros((G, N, 4, 4), np.float32),
        dense_JtJ=np.zeros((G, dim, dim), np.float32), dense_Jtr=np.zeros((G, dim), np.float32),
        rhs=np.zeros((G, N, 6), np.float32), precond=np.zeros((G, N, 6), np.float32),
        pcg_scalars=np.zeros((G, L, 4), np.float32), dense_count=np.zeros((G, max(P, 1)), np.int32),
        delta=np.zeros((G, N, 6), np.float32),
    )
    ct = OrcTrace(*[_p(getattr(t, f[0])) for f in OrcTrace._fields_]) if want_trace else None
    rc = lib().orc_solve(C.byref(prm), N, Wd, Hd, _p(intr), _p(campos), _p(normals), _p(corr), int(corr.shape[0]),
                         _p(pairs), P, _p(poses_io), C.byref(ct) if ct is not None else None)
    if rc != 0:
        raise ValueError(f"orc_solve failed rc={rc}")
    t.poses = poses_io.reshape(N, 4, 4)
    return t


def solve_jobs(instances, n_jobs, n_workers, params: OrcParams | None = None) -> int:
    """n_jobs whole solves over `instances` (dicts with campos, normals, intr, corr, poses; job j takes instance j % len) on
    n_workers OpenMP threads, one single-threaded solve each (orc_solve_jobs).  Returns the number of solves completed."""
    prm = params or default_params()
    B = len(instances)
    cam = [np.ascontiguousarray(q["campos"], np.float32) for q in instances]
    nrm = [np.ascontiguousarray(q["normals"], np.float32) for q in instances]
    cor = [np.ascontiguousarray(q["corr"], ENTRYJ_DTYPE) for q in instances]
    pos = [np.ascontiguousarray(q["poses"], np.float32).reshape(-1, 16) for q in instances]
    N, Hd, Wd = cam[0].shape[:3]
    intr = np.ascontiguousarray(np.stack([q["intr"] for q in instances]), np.float32)
    pairs = np.ascontiguousarray(target_lower_pairs(N), np.int32).reshape(-1, 2)
    arr = lambda xs: (C.c_void_p * B)(*[x.ctypes.data for x in xs])
    nC = np.array([c.shape[0] for c in cor], np.int32)
    f = lib().orc_solve_jobs
    f.restype = C.c_int
    return int(f(C.byref(prm), int(n_jobs), int(n_workers), B, N, Wd, Hd, _p(intr), arr(cam), arr(nrm), arr(cor), _p(nC), _p(pairs), pairs.shape[0], arr(pos)))


def sparse_apply(corr, T, p, params: OrcParams | None = None) -> np.ndarray:
    """Matrix-free sparse J^T J p exactly as PCGStep_Kernel0 + PCGStep_Kernel1a apply it."""
    prm = params or default_params()
    corr = np.ascontiguousarray(corr, ENTRYJ_DTYPE)
    T = np.ascontiguousarray(T, np.float32)
    N = T.shape[0]
    p = np.ascontiguousarray(p, np.float32).reshape(N, 6)
    out = np.zeros((N, 6), np.float32)
    lib().orc_sparse_apply(C.byref(prm), N, _p(corr), int(corr.shape[0]), _p(T.reshape(N, 16)), _p(p), _p(out))
    return out


# ---- depth pre-processing and normals -------------------------------------------
def erode_depth(depth, radius=1, diff=0.001, ratio=0.8) -> np.ndarray:
    d = np.ascontiguousarray(depth, np.float32); H, W = d.shape
    out = np.zeros_like(d)
    lib().orc_erode_depth(_p(d), _p(out), W, H, int(radius), C.c_float(diff), C.c_float(ratio))
    return out


def gauss_filter_depth(depth, radius=2, sigma_d=2.0, sigma_r=100000.0) -> np.ndarray:
    d = np.ascontiguousarray(depth, np.float32); H, W = d.shape
    out = np.zeros_like(d)
    lib().orc_gauss_filter_depth(_p(d), _p(out), W, H, int(radius), C.c_float(sigma_d), C.c_float(sigma_r))
    return out


def process_depth(depth, erode_radius=1, erode_diff=0.001, erode_ratio=0.8, bf_radius=2, sigma_d=2.0, sigma_r=100000.0) -> np.ndarray:
    d = np.ascontiguousarray(depth, np.float32); H, W = d.shape
    out = np.zeros_like(d)
    lib().orc_process_depth(_p(d), _p(out), W, H, int(erode_radius), C.c_float(erode_diff), C.c_float(erode_ratio), int(bf_radius), C.c_float(sigma_d), C.c_float(sigma_r))
    return out


def depth_to_normals(depth, K):
    """Returns (normals [H,W,4], xyz [H,W,4]).  K^-1 through the same generic cofactor inverse as the device math."""
    d = np.ascontiguousarray(depth, np.float32); H, W = d.shape
    K4 = np.eye(4, dtype=np.float32); K4[:3, :3] = np.asarray(K, np.float32)
    Ki = np.ascontiguousarray(mat4_inverse(K4), np.float32).reshape(16)
    normals = np.zeros((H, W, 4), np.float32); xyz = np.zeros((H, W, 4), np.float32)
    lib().orc_depth_to_normals(_p(d), W, H, _p(Ki), _p(normals), _p(xyz))
    return normals, xyz


# ---- correspondence RANSAC (oracle/btba_oracle_ransac.c) ------------------------------------------------

def _pts4(p):
    p = np.asarray(p, np.float32)
    if p.ndim == 2 and p.shape[1] == 3:
        p = np.concatenate([p, np.ones((p.shape[0], 1), np.float32)], 1)
    return np.ascontiguousarray(p, np.float32)


def procrustes(src, dst):
    """procrustesKernel (cuda_ransac.cu:999-1102): (ok, pose[4,4], gap)."""
    s, d = _pts4(src), _pts4(dst)
    pose = np.zeros(16, np.float32)
    gap = C.c_float(0)
    f = lib().orc_procrustes
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    ok = f(s.ctypes.data, d.ctypes.data, s.shape[0], pose.ctypes.data, C.byref(gap))
    return bool(ok), pose.reshape(4, 4), float(gap.value)


def ransac_draw(seed, pair, trial, draw, n_pts):
    f = lib().orc_ransac_draw
    f.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    f.restype = C.c_int32
    return int(f(seed, pair, trial, draw, n_pts))


def procrustes_reference(src, dst):
    """procrustesKernel as the reference runs it -- with its approximate 3x3 SVD, restated operation for operation
    (orc_procrustes_reference): (ok, pose [4,4]); ok False = the reference's "R is not valid"."""
    s, d = _pts4(src), _pts4(dst)
    pose = np.zeros(16, np.float32)
    f = lib().orc_procrustes_reference
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    ok = f(s.ctypes.data, d.ctypes.data, s.shape[0], pose.ctypes.data)
    return bool(ok), pose.reshape(4, 4)


def ransac_pair(ptsA, ptsB, n_trials, dist_thres, samples=None, seed=0, pair_id=0, hypothesis=1):
    """One frame pair of ransacMultiPairGPU.  hypothesis 0: the reference's approximate-SVD procrustes; 1 (default here): the exact
    Kabsch optimum with the collinearity gap.  Returns dict(inlier_ids, best_trial, best_pose, counts, poses)."""
    a, b = _pts4(ptsA), _pts4(ptsB)
    n = a.shape[0]
    ids = np.zeros(max(n, 1), np.int32)
    n_in, best = C.c_int32(0), C.c_int32(-1)
    bp = np.zeros(16, np.float32)
    counts = np.zeros(n_trials, np.int32)
    poses = np.zeros((n_trials, 16), np.float32)
    smp = None if samples is None else np.ascontiguousarray(samples, np.int32).reshape(n_trials, 3)
    f = lib().orc_ransac_pair_ex
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_uint64, C.c_int,
                  C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p]
    f(int(hypothesis), a.ctypes.data, b.ctypes.data, n, n_trials, dist_thres, smp.ctypes.data if smp is not None else None, seed, pair_id,
      ids.ctypes.data, C.byref(n_in), C.byref(best), bp.ctypes.data, counts.ctypes.data, poses.ctypes.data)
    return dict(inlier_ids=ids[: n_in.value].copy(), best_trial=int(best.value), best_pose=bp.reshape(4, 4), counts=counts,
                poses=poses.reshape(n_trials, 4, 4))


# ---- cuRAND XORWOW as the reference's RANSAC draws from it (oracle/xorwow.h) -------------------------------------

def curand_xorwow_state(seed: int, subsequence: int, offset: int = 0) -> np.ndarray:
    """curand_init(seed, subsequence, offset): uint32 [6] = d, v[0..4]."""
    out = np.zeros(6, np.uint32)
    f = lib().orc_curand_xorwow_state
    f.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    f.restype = None
    f(seed, subsequence, offset, out.ctypes.data)
    return out


def curand_xorwow_draw(seed: int, subsequence: int, offset: int, n: int):
    """n draws after curand_init(seed, subsequence, offset): (raw uint32 [n] = curand(), float32 [n] = curand_uniform())."""
    raw, u = np.zeros(n, np.uint32), np.zeros(n, np.float32)
    f = lib().orc_curand_xorwow_draw
    f.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    f.restype = None
    f(seed, subsequence, offset, n, raw.ctypes.data, u.ctypes.data)
    return raw, u


def ransac_reference_uniforms(n_trials: int, seed: int = 0) -> np.ndarray:
    """The reference's per-trial uniforms (curand_init(seed, trial, 0), three curand_uniform): float32 [n_trials, 3]."""
    u = np.zeros((n_trials, 3), np.float32)
    f = lib().orc_ransac_reference_uniforms
    f.argtypes = [C.c_uint64, C.c_int, C.c_void_p]
    f.restype = None
    f(seed, n_trials, u.ctypes.data)
    return u


def ransac_reference_samples(n_trials: int, n_pts: int, seed: int = 0) -> np.ndarray:
    """round(curand_uniform * (n_pts - 1)) x 3 per trial (cuda_ransac.cu:1154-1161): int32 [n_trials, 3]."""
    s = np.zeros((n_trials, 3), np.int32)
    f = lib().orc_ransac_reference_samples
    f.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_void_p]
    f.restype = None
    f(seed, n_trials, n_pts, s.ctypes.data)
    return s


def xorwow_matrix(which: int, k: int) -> np.ndarray:
    """A^(2^k) (which 0) or A^(2^(67+k)) (which 1) of the XORWOW recurrence over GF(2): uint32 [160, 5], row c = image of unit bit c."""
    m = np.zeros((160, 5), np.uint32)
    f = lib().orc_xorwow_matrix
    f.argtypes = [C.c_int, C.c_int, C.c_void_p]
    f.restype = None
    f(which, k, m.ctypes.data)
    return m


# ---- keyframe memory (oracle/btba_oracle_keyframes.c: Bundler.cpp:185-274, Utils.cpp:42-47; parity unpinned, see its header) ----
def rotation_geodesic_distance(pose1, pose2) -> float:
    f = lib().orc_rotation_geodesic_distance
    f.argtypes = [C.c_void_p, C.c_void_p]
    f.restype = C.c_float
    a, b = np.ascontiguousarray(pose1, np.float32).reshape(16), np.ascontiguousarray(pose2, np.float32).reshape(16)
    return float(f(_p(a), _p(b)))


def keyframe_pool(poses, ids=None, status_other=None, n_keypts=None, min_feat_num=0, min_rot_deg=10.0):
    """checkAndAddKeyframe over frames 0 .. M-1 in order: (added flags [M] bool, keyframe indices)."""
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    M = len(poses)
    f = lib().orc_check_and_add_keyframe
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    f.restype = C.c_int
    kf, n = np.zeros(M + 1, np.int32), np.zeros(1, np.int32)
    added = np.zeros(M, bool)
    for k in range(M):
        added[k] = bool(f(_p(poses), k, k if ids is None else int(ids[k]), 1 if status_other is None else int(status_other[k]),
                          100 if n_keypts is None else int(n_keypts[k]), min_feat_num, min_rot_deg, _p(kf), _p(n)))
    return added, kf[: n[0]].copy()


def select_keyframes_for_ba(poses, newframe, keyframes, max_BA_frames, addr_rank=None):
    """selectKeyFramesForBA: the chosen frame indices in the reference's set order (addr_rank: rank of every frame's address, None: index order)."""
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    keyframes = np.ascontiguousarray(keyframes, np.int32)
    f = lib().orc_select_keyframes_for_ba
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    f.restype = C.c_int
    out = np.zeros(len(keyframes) + 2, np.int32)
    rank = None if addr_rank is None else np.ascontiguousarray(addr_rank, np.int32)
    n = f(_p(poses), int(newframe), _p(keyframes), len(keyframes), int(max_BA_frames), None if rank is None else _p(rank), _p(out))
    return out[:n].copy()
