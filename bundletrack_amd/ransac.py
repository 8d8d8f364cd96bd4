"""Correspondence RANSAC on the MI355X (btba_ransac_pairs) and the caller logic around it.

Mirrors SiftManager::runRansacMultiPairGPU (src/FeatureManager.cpp:659-741) -> ransacMultiPairGPU
(src/cuda/cuda_ransac.cu:1228-1323): matches are moved into the model frame with the frames' current poses, every
frame pair votes n_trials 3-point rigid hypotheses, the inliers of the best one survive, and a pair left with fewer
than 5 matches loses all of them."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib


def _pts4(p) -> np.ndarray:
    p = np.asarray(p, np.float32)
    if p.ndim != 2 or p.shape[1] not in (3, 4):
        raise ValueError("points must be [n,3] or [n,4]")
    if p.shape[1] == 3:
        p = np.concatenate([p, np.ones((p.shape[0], 1), np.float32)], 1)
    return np.ascontiguousarray(p, np.float32)


def pack_points(ptsA, ptsB):
    """Per-pair point lists -> the C ABI's layout: (float4 A points back to back, float4 B points, int32 n_pts[n_pairs])."""
    if len(ptsA) != len(ptsB) or len(ptsA) == 0:
        raise ValueError("need the same, non-zero number of point sets on both sides")
    A = [_pts4(a) for a in ptsA]
    B = [_pts4(b) for b in ptsB]
    if any(a.shape != b.shape for a, b in zip(A, B)):
        raise ValueError("a pair's two point sets must have the same length")
    n_pts = np.array([a.shape[0] for a in A], np.int32)
    T = int(n_pts.sum())
    a_all = np.concatenate(A) if T else np.zeros((1, 4), np.float32)
    b_all = np.concatenate(B) if T else np.zeros((1, 4), np.float32)
    return a_all, b_all, n_pts


def reference_uniforms(n_trials: int, seed: int = 0) -> np.ndarray:
    """btba_ransac_reference_uniforms: the reference's sample stream as float32 [n_trials, 3] -- row t holds the three
    curand_uniform() draws after curand_init(seed, t, 0) (cuda_ransac.cu:1154-1161).  Host-only, no GPU needed."""
    u = np.zeros((max(int(n_trials), 0), 3), np.float32)
    f = lib().btba_ransac_reference_uniforms
    f.argtypes = [C.c_uint64, C.c_int, C.c_void_p]
    check(f(int(seed), int(n_trials), u.ctypes.data), "btba_ransac_reference_uniforms")
    return u


def reference_samples(n_trials: int, n_pts: int, seed: int = 0) -> np.ndarray:
    """The triples the reference's trial t draws on a pair of n_pts points: round(u * (n_pts - 1)) in fp32, int32 [n_trials, 3]."""
    prod = (reference_uniforms(n_trials, seed) * np.float32(n_pts - 1)).astype(np.float64)       # the fp32 product, then exactly:
    return (np.sign(prod) * np.floor(np.abs(prod) + 0.5)).astype(np.int32)                        # roundf, half away from zero


def ransac_packed(ws, a_all, b_all, n_pts, n_trials: int = 2000, inlier_dist: float = 0.01, samples=None, seed: int = 0,
                  want_trials: bool = False, hypothesis: int = 0) -> list[dict]:
    """btba_ransac_pairs_ex on already packed points (see pack_points).  hypothesis: _lib.RANSAC_REFERENCE_SVD (0, default: the
    reference's procrustesKernel with its approximate 3x3 SVD, operation for operation) or _lib.RANSAC_HORN (1: exact Kabsch),
    optionally ORed with _lib.RANSAC_DRAW_HASH.  samples None: the reference's cuRAND XORWOW triples for curand_init(seed, trial, 0)
    (the reference's seed is 0), or with RANSAC_DRAW_HASH a per-pair counter hash of the seed."""
    n_pairs, T = len(n_pts), int(np.sum(n_pts))
    smp = None
    if samples is not None:
        smp = np.ascontiguousarray(samples, np.int32)
        if smp.shape != (n_pairs, n_trials, 3):
            raise ValueError("samples must be int32 [n_pairs, n_trials, 3]")
    ids = np.zeros(max(T, 1), np.int32)
    n_in = np.zeros(n_pairs, np.int32)
    best = np.zeros(n_pairs, np.int32)
    pose = np.zeros((n_pairs, 16), np.float32)
    counts = np.zeros((n_pairs, n_trials), np.int32) if want_trials else None
    poses = np.zeros((n_pairs, n_trials, 12), np.float32) if want_trials else None
    f = lib().btba_ransac_pairs_ex
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_uint64,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    check(f(ws.handle, int(hypothesis), 0, n_pairs, a_all.ctypes.data, b_all.ctypes.data, n_pts.ctypes.data, int(n_trials), float(inlier_dist),
            smp.ctypes.data if smp is not None else None, int(seed), ids.ctypes.data, n_in.ctypes.data, best.ctypes.data,
            pose.ctypes.data, counts.ctypes.data if want_trials else None, poses.ctypes.data if want_trials else None),
          "btba_ransac_pairs")
    out, o = [], 0
    for p in range(n_pairs):
        r = dict(inlier_ids=ids[o:o + n_in[p]].copy(), best_trial=int(best[p]), best_pose=pose[p].reshape(4, 4).copy())
        if want_trials:
            r["counts"], r["poses"] = counts[p], poses[p].reshape(n_trials, 3, 4)
        out.append(r)
        o += int(n_pts[p])
    return out


def ransac_packed_device(ws, a_dev, b_dev, n_pts, n_trials: int = 2000, inlier_dist: float = 0.01, samples_dev=None, seed: int = 0, hypothesis: int = 0):
    """Device-resident form (btba_ransac_pairs_ex, device_resident = 1): a_dev / b_dev float32 CUDA tensors [T, 4] of all pairs'
    points back to back, n_pts host int32 [n_pairs].  Returns CUDA tensors (inlier_ids [T], n_inliers [n_pairs], best_trial
    [n_pairs], best_pose [n_pairs, 16]); asynchronous on the workspace stream -- no point or result crosses PCIe."""
    import torch
    from .optimizer import _dev_ptr
    n_pts = np.ascontiguousarray(n_pts, np.int32)
    n_pairs, T = len(n_pts), int(n_pts.sum())
    dev = a_dev.device
    ids = torch.zeros(max(T, 1), dtype=torch.int32, device=dev)
    n_in = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
    best = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
    pose = torch.zeros((n_pairs, 16), dtype=torch.float32, device=dev)
    f = lib().btba_ransac_pairs_ex
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_uint64,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    check(f(ws.handle, int(hypothesis), 1, n_pairs, _dev_ptr(a_dev, "ptsA"), _dev_ptr(b_dev, "ptsB"), n_pts.ctypes.data, int(n_trials), float(inlier_dist),
            _dev_ptr(samples_dev, "samples") if samples_dev is not None else None, int(seed), _dev_ptr(ids, "ids"), _dev_ptr(n_in, "n_in"), _dev_ptr(best, "best"),
            _dev_ptr(pose, "pose"), None, None), "btba_ransac_pairs_ex")
    return ids, n_in, best, pose


def ransac_multi_pair(ws, ptsA, ptsB, n_trials: int = 2000, inlier_dist: float = 0.01, samples=None, seed: int = 0,
                      want_trials: bool = False, hypothesis: int = 0) -> list[dict]:
    """ptsA[p], ptsB[p]: [n_p,3|4] model-frame points of frame pair p (A is moved onto B).  samples: optional
    int32 [n_pairs, n_trials, 3].  Returns one dict per pair: inlier_ids (ascending), best_trial (-1: none),
    best_pose [4,4], and with want_trials also counts [n_trials] and poses [n_trials,3,4]."""
    a_all, b_all, n_pts = pack_points(ptsA, ptsB)
    return ransac_packed(ws, a_all, b_all, n_pts, n_trials, inlier_dist, samples, seed, want_trials, hypothesis)


def run_ransac_multi_pair(ws, pairs, matches: dict, n_trials: int = 2000, inlier_dist: float = 0.01, seed: int = 0, hypothesis: int = 0) -> None:
    """SiftManager::runRansacMultiPairGPU.  pairs: [(frameA, frameB)] (A newer); matches[(A.id, B.id)] = (ptA_cam,
    ptB_cam) is replaced IN PLACE by its RANSAC inliers, or emptied when fewer than 5 survive (:733-737).  Defaults = the
    reference: procrustesKernel hypotheses on the cuRAND XORWOW triples of curand_init(0, trial, 0)."""
    keys, A, B = [], [], []
    for fa, fb in pairs:
        key = (fa.id, fb.id)
        pa, pb = matches.get(key, (np.zeros((0, 3), np.float32),) * 2)
        Ta, Tb = np.asarray(fa.pose_in_model, np.float32), np.asarray(fb.pose_in_model, np.float32)
        keys.append(key)
        A.append(np.asarray(pa, np.float32) @ Ta[:3, :3].T + Ta[:3, 3])          # pcl::transformPointWithNormal, fp32
        B.append(np.asarray(pb, np.float32) @ Tb[:3, :3].T + Tb[:3, 3])
    if not keys:
        return
    res = ransac_multi_pair(ws, A, B, n_trials=n_trials, inlier_dist=inlier_dist, seed=seed, hypothesis=hypothesis)
    for key, r in zip(keys, res):
        pa, pb = matches.get(key, (np.zeros((0, 3), np.float32),) * 2)
        keep = r["inlier_ids"]
        if len(keep) < 5:
            matches[key] = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
        else:
            matches[key] = (np.asarray(pa, np.float32)[keep], np.asarray(pb, np.float32)[keep])
