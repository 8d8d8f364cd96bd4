"""Seeded synthetic bundle-adjustment problems.

The reference repo ships no dataset, masks or features, so every
test and benchmark input is generated here: an ellipsoid object seen by K RGB-D keyframes on
an orbit, analytic depth + normals in the formats `Frame` hands to the optimiser
(src/Frame.h:73-75: depth float[H*W] metres with 0 = invalid, normals float4[H*W] with
w = 0 and (0,0,0,0) = invalid), perturbed initial poses and noisy 3D-3D feature
correspondences in the EntryJ wire format (src/cuda/SIFTImageManager.h:44-59).
numpy only -- no GPU, no torch.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

ENTRYJ_DTYPE = np.dtype(
    [("imgIdx_i", "<u4"), ("imgIdx_j", "<u4"), ("pos_i", "<f4", (3,)), ("pos_j", "<f4", (3,))]
)

# NOCS intrinsics, src/DataLoader.cpp:75-77
NOCS_K = np.array([[591.0125, 0.0, 322.525], [0.0, 590.16775, 244.11084], [0.0, 0.0, 1.0]], np.float64)
SEMI_AXES = np.array([0.06, 0.09, 0.05])
OBJ_DIST = 0.7
BACKGROUND_RADIUS = 1.0   # inner sphere shell around the model origin = 0.3 m behind the object


def so3_exp(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th**2 * Kx @ Kx


def se3_exp(w: np.ndarray, u: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = so3_exp(w)
    if th < 1e-12:
        V = np.eye(3) + 0.5 * Kx
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * Kx + (th - np.sin(th)) / th**3 * Kx @ Kx
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ u
    return T


def orbit_pose(angle_rad: float, tilt_rad: float = 0.35) -> np.ndarray:
    """Camera->model transform of a camera on an orbit of radius OBJ_DIST looking at the origin."""
    c = OBJ_DIST * np.array([np.sin(angle_rad) * np.cos(tilt_rad), -np.sin(tilt_rad), -np.cos(angle_rad) * np.cos(tilt_rad)])
    z = -c / np.linalg.norm(c)                    # optical axis towards the origin
    up = np.array([0.0, -1.0, 0.0])
    x = np.cross(-up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, c
    return T


def render(T_cam2model: np.ndarray, K: np.ndarray, xs: np.ndarray, ys: np.ndarray, background: bool):
    """Analytic depth (z in camera frame) and camera-facing unit normals at pixel coords xs, ys.
    Returns depth [...] f32 and normals [...,4] f32 (w = 0; invalid = zeros)."""
    shape = xs.shape
    xs = xs.reshape(-1).astype(np.float64)
    ys = ys.reshape(-1).astype(np.float64)
    d_cam = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], 1)
    R, o = T_cam2model[:3, :3], T_cam2model[:3, 3]
    d = d_cam @ R.T                                # ray directions in model frame (z_cam component = 1)
    depth = np.zeros(xs.shape[0])
    nrm_m = np.zeros((xs.shape[0], 3))
    # ellipsoid
    A = 1.0 / SEMI_AXES**2
    a = (d * d * A).sum(1)
    b = 2 * (d * o * A).sum(1)
    c = (o * o * A).sum() - 1.0
    disc = b * b - 4 * a * c
    hit = disc > 0
    t = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0))) / (2 * a), 0.0)
    hit &= t > 0
    p = o + t[:, None] * d
    n = p * A
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    depth[hit] = t[hit]
    nrm_m[hit] = n[hit]
    if background:
        a2 = (d * d).sum(1)
        b2 = 2 * (d @ o)
        c2 = o @ o - BACKGROUND_RADIUS**2
        t2 = (-b2 + np.sqrt(b2 * b2 - 4 * a2 * c2)) / (2 * a2)   # far intersection (camera is inside)
        p2 = o + t2[:, None] * d
        n2 = -p2 / np.linalg.norm(p2, axis=1, keepdims=True)       # inward = towards the camera
        miss = ~hit
        depth[miss] = t2[miss]
        nrm_m[miss] = n2[miss]
    nrm_c = nrm_m @ R                               # model -> camera: R^T n
    # orient to camera (already true by construction; enforce)
    flip = (nrm_c * d_cam).sum(1) > 0
    nrm_c[flip] *= -1
    normals = np.zeros((xs.shape[0], 4), np.float32)
    normals[:, :3] = nrm_c
    normals[depth == 0] = 0
    return depth.astype(np.float32).reshape(shape), normals.reshape(shape + (4,))


@dataclass
class Problem:
    """One bundle-adjustment call: what Bundler::optimizeGPU hands to OptimizerGpu::optimizeFrames."""
    K: np.ndarray                 # [3,3] f32 full-resolution intrinsics
    H: int
    W: int
    depth: np.ndarray | None      # [N,H,W] f32 (None when only the cached resolution was rendered)
    normals: np.ndarray | None    # [N,H,W,4] f32
    corr: np.ndarray              # ENTRYJ_DTYPE[C], pair-major (outer i, inner j>i)
    n_match_per_pair: np.ndarray  # [P] int32 in the same order
    poses_init: np.ndarray        # [N,4,4] f32 camera->model initial estimates (frame 0 exact)
    poses_gt: np.ndarray          # [N,4,4] f64
    cache_depth: np.ndarray | None = None    # [N,Hd,Wd] depth sampled at the cache's source pixels
    cache_normals: np.ndarray | None = None  # [N,Hd,Wd,4]
    downscale: int = 4

    @property
    def n_frames(self) -> int:
        return self.poses_init.shape[0]


def cache_source_pixels(H: int, W: int, Hd: int, Wd: int):
    """Full-res pixel picked for every downsampled pixel, CUDAImageUtil.cu:57-61 (fp32 arithmetic)."""
    sw = np.float32(W - 1) / np.float32(Wd - 1)
    sh = np.float32(H - 1) / np.float32(Hd - 1)
    xi = (np.arange(Wd, dtype=np.float32) * sw + np.float32(0.5)).astype(np.uint32)
    yi = (np.arange(Hd, dtype=np.float32) * sh + np.float32(0.5)).astype(np.uint32)
    return xi, yi


def _sample_surface(rng: np.random.Generator, n: int) -> tuple[np.ndarray, np.ndarray]:
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    p = v * SEMI_AXES
    nrm = p / SEMI_AXES**2
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return p, nrm


def make_problem(n_frames: int, corr_per_pair: int, seed: int, *, background: bool = True,
                 H: int = 480, W: int = 640, downscale: int = 4, full_res: bool = True,
                 rot_step_deg=(10.0, 12.0), perturb_deg: float = 2.0, perturb_m: float = 0.005,
                 noise_m: float = 0.001, outlier_frac: float = 0.05, K: np.ndarray | None = None,
                 angles: np.ndarray | None = None) -> Problem:
    """orbit of `n_frames` keyframes (consecutive rotation 10-12 deg), GT
    poses perturbed by U(+-2 deg, +-5 mm) (frame 0 exact), `corr_per_pair` surface points per
    frame pair (+N(0, 1 mm), 5 % outliers displaced 2-5 cm, shuffled inside the pair)."""
    rng = np.random.default_rng(seed)
    K = NOCS_K if K is None else np.asarray(K, np.float64)
    steps = np.deg2rad(rng.uniform(rot_step_deg[0], rot_step_deg[1], size=n_frames - 1))
    if angles is None:
        angles = np.concatenate([[0.0], np.cumsum(steps)])
    assert len(angles) == n_frames
    poses_gt = np.stack([orbit_pose(a) for a in angles])
    # model frame = frame 0's view of the object is arbitrary; keep object at the model origin.
    poses_init = poses_gt.copy()
    for k in range(1, n_frames):
        w = np.deg2rad(rng.uniform(-perturb_deg, perturb_deg, 3))
        u = rng.uniform(-perturb_m, perturb_m, 3)
        poses_init[k] = poses_gt[k] @ se3_exp(w, u)

    Hd, Wd = int(H / downscale), int(W / downscale)
    depth = normals = None
    if full_res:
        ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        depth = np.zeros((n_frames, H, W), np.float32)
        normals = np.zeros((n_frames, H, W, 4), np.float32)
        for k in range(n_frames):
            depth[k], normals[k] = render(poses_gt[k], K, xs, ys, background)
        xi, yi = cache_source_pixels(H, W, Hd, Wd)
        cache_depth = depth[:, yi][:, :, xi]
        cache_normals = normals[:, yi][:, :, xi]
    else:
        xi, yi = cache_source_pixels(H, W, Hd, Wd)
        ys, xs = np.meshgrid(yi, xi, indexing="ij")
        cache_depth = np.zeros((n_frames, Hd, Wd), np.float32)
        cache_normals = np.zeros((n_frames, Hd, Wd, 4), np.float32)
        for k in range(n_frames):
            cache_depth[k], cache_normals[k] = render(poses_gt[k], K, xs, ys, background)

    # correspondences, pair-major like Bundler::optimizeGPU (src/Bundler.cpp:298-324)
    inv = np.linalg.inv(poses_gt)
    cam_centres = poses_gt[:, :3, 3]
    corr_blocks, counts = [], []
    for i in range(n_frames):
        for j in range(i + 1, n_frames):
            m = corr_per_pair
            pts = np.zeros((0, 3))
            tries = 0
            while pts.shape[0] < m and tries < 6:
                cand, nrm = _sample_surface(rng, max(4 * m, 4096))
                vis = (((cam_centres[i] - cand) * nrm).sum(1) > 0.02) & (((cam_centres[j] - cand) * nrm).sum(1) > 0.02)
                pts = np.concatenate([pts, cand[vis]])
                tries += 1
            if pts.shape[0] < m:           # no common visible region: 3D-3D residuals do not need one
                extra, _ = _sample_surface(rng, m - pts.shape[0])
                pts = np.concatenate([pts, extra])
            pts = pts[:m]
            pi = pts @ inv[i, :3, :3].T + inv[i, :3, 3] + rng.normal(scale=noise_m, size=(m, 3))
            pj = pts @ inv[j, :3, :3].T + inv[j, :3, 3] + rng.normal(scale=noise_m, size=(m, 3))
            n_out = int(round(outlier_frac * m))
            if n_out:
                idx = rng.choice(m, n_out, replace=False)
                dirs = rng.normal(size=(n_out, 3))
                dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
                pj[idx] += dirs * rng.uniform(0.02, 0.05, size=(n_out, 1))
            perm = rng.permutation(m)
            blk = np.zeros(m, ENTRYJ_DTYPE)
            blk["imgIdx_i"], blk["imgIdx_j"] = i, j
            blk["pos_i"], blk["pos_j"] = pi[perm], pj[perm]
            corr_blocks.append(blk)
            counts.append(m)
    corr = np.concatenate(corr_blocks) if corr_blocks else np.zeros(0, ENTRYJ_DTYPE)
    return Problem(K=K.astype(np.float32), H=H, W=W, depth=depth, normals=normals, corr=corr,
                   n_match_per_pair=np.asarray(counts, np.int32), poses_init=poses_init.astype(np.float32),
                   poses_gt=poses_gt, cache_depth=cache_depth, cache_normals=cache_normals, downscale=downscale)


def pruned_pool_angles(pool_size: int, keep: int, seed: int, rot_step_deg=(10.0, 12.0)) -> np.ndarray:
    """a pool of `pool_size` keyframes on the orbit plus one new frame, pruned to `keep`
    frames by the reference's greedy-rotation subset selection (Bundler.cpp:222-274); returns their orbit angles."""
    from .bundler import FrameRef, KeyframeMemory
    rng = np.random.default_rng(seed)
    steps = np.deg2rad(rng.uniform(rot_step_deg[0], rot_step_deg[1], size=pool_size))
    ang = np.concatenate([[0.0], np.cumsum(steps)])
    frames = [FrameRef(id=k, pose_in_model=orbit_pose(a).astype(np.float32)) for k, a in enumerate(ang)]
    mem = KeyframeMemory(max_BA_frames=keep, keyframes=frames[:pool_size])
    chosen = mem.select_keyframes_for_ba(frames[pool_size])
    return np.array([ang[f.id] for f in chosen])


def config_seed(config: int, instance: int = 0) -> int:
    """Seeds `1234 + 1000*config + instance`."""
    return 1234 + 1000 * config + instance


def rotation_angle(Ra: np.ndarray, Rb: np.ndarray) -> float:
    """Geodesic angle between two rotations.  atan2 of the skew part's norm and the trace
    term, so that fp32-orthonormality noise (1-cos ~ 1e-8) is not amplified the way acos is."""
    D = Ra @ Rb.T
    s = 0.5 * np.linalg.norm([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
    c = (np.trace(D) - 1) / 2
    return float(np.arctan2(s, c))


def pose_error(Ta: np.ndarray, Tb: np.ndarray) -> tuple[float, float]:
    """(rotation angle [rad], translation distance [m]) between two 4x4 poses."""
    Ta = np.asarray(Ta, np.float64)
    Tb = np.asarray(Tb, np.float64)
    return rotation_angle(Ta[:3, :3], Tb[:3, :3]), float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))


def analytic_cache(pb: Problem):
    """Frame cache (camPos float4, normals float4, downscaled intrinsics) computed on the host from the
    depth/normals rendered at the cache's source pixels -- for benchmarks and property tests that do not
    need the full-resolution frames.  Same formulas as CUDACache (CUDACache.cpp:20-24,
    CUDAImageUtil.cu:310-327) in fp32; not guaranteed bit-identical to the device cache build."""
    Hd, Wd = pb.cache_depth.shape[1:]
    xi, yi = cache_source_pixels(pb.H, pb.W, Hd, Wd)
    K = pb.K.astype(np.float32)
    d = pb.cache_depth.astype(np.float32)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x = (xi[None, None, :].astype(np.float32) * d) / fx - (cx / fx) * d
    y = (yi[None, :, None].astype(np.float32) * d) / fy - (cy / fy) * d
    campos = np.stack([x, y, d, np.ones_like(d)], -1).astype(np.float32)
    campos[d < 0.1] = 0
    intr = np.array([fx * (np.float32(Wd) / np.float32(pb.W)), fy * (np.float32(Hd) / np.float32(pb.H)),
                     cx * (np.float32(Wd - 1) / np.float32(pb.W - 1)), cy * (np.float32(Hd - 1) / np.float32(pb.H - 1))], np.float32)
    return campos, pb.cache_normals.astype(np.float32), intr


def compact_cache(pb: Problem):
    """Compact (z, nx, ny, nz) frame cache [N, Hd, Wd, 4] built on the host (include/btba.h, "ZN"): the z lane is the GATED
    depth -- 0 where the reference's camPos is zero (d < 0.1 or NaN), like btba_build_cache_zn writes it."""
    d = pb.cache_depth.astype(np.float32)
    with np.errstate(invalid="ignore"):
        d = np.where(d >= np.float32(0.1), d, np.float32(0.0)).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([d[..., None], pb.cache_normals[..., :3].astype(np.float32)], -1), np.float32)


# ---------------------------------------------------------------------------------------------------------
# A synthetic tracking SEQUENCE: frames on an orbit handed one by one to
# bundler.Bundler, with a stand-in for the feature front end (LF-Net / matching / RANSAC are out of scope).
# ---------------------------------------------------------------------------------------------------------

class SyntheticSequence:
    """`n_frames` views of the ellipsoid on an orbit (`step_deg` per frame), rendered on demand."""

    def __init__(self, n_frames: int = 60, seed: int = 2234, *, step_deg=(5.5, 6.5), background: bool = False,
                 H: int = 480, W: int = 640, K: np.ndarray | None = None):
        rng = np.random.default_rng(seed)
        self.seed, self.n_frames, self.H, self.W, self.background = seed, n_frames, H, W, background
        self.K = (NOCS_K if K is None else np.asarray(K, np.float64))
        steps = np.deg2rad(rng.uniform(step_deg[0], step_deg[1], size=n_frames - 1))
        self.poses_gt = np.stack([orbit_pose(a) for a in np.concatenate([[0.0], np.cumsum(steps)])])   # camera -> model
        self._ys, self._xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")

    def render(self, k: int):
        """(depth [H,W] f32, normals [H,W,4] f32) of frame k, masked to the object unless `background`."""
        return render(self.poses_gt[k], self.K, self._xs, self._ys, self.background)


class SyntheticFeatureManager:
    """Stand-in for the slice of SiftManager that Bundler calls (FeatureManager.h:86-130): matches are surface
    points seen by both frames, expressed in each camera frame (+ N(0, noise_m), `outlier_frac` displaced 2-5 cm),
    deterministic per frame pair.  `procrustes_by_correspondence` is the reference's Kabsch on the matches moved
    into the model frame with the frames' CURRENT poses (FeatureManager.cpp:523-556); the outliers the reference
    would have pruned by RANSAC beforehand are pruned here with one residual gate."""

    def __init__(self, seq: SyntheticSequence, corr_per_pair: int = 300, *, noise_m: float = 0.001, outlier_frac: float = 0.05,
                 inlier_dist: float = 0.01, ransac=None):
        self.seq, self.m, self.noise_m, self.outlier_frac, self.inlier_dist = seq, corr_per_pair, noise_m, outlier_frac, inlier_dist
        self.ransac = ransac              # optional callable(pairs, matches): SiftManager::findCorres ends in runRansacBetween (:191)
        self.matches: dict = {}
        self._inv = np.linalg.inv(seq.poses_gt)
        self.gt_index: dict = {}          # frame id -> index into the sequence (ids are re-assigned by Bundler)

    def register(self, frame, seq_index: int) -> None:
        self.gt_index[id(frame)] = seq_index

    def _gt(self, frame) -> int:
        return self.gt_index[id(frame)]

    def forget_frame(self, frame) -> None:
        for key in [k for k in self.matches if frame.id in k]:
            del self.matches[key]

    def find_corres(self, frameA, frameB) -> None:
        key = (frameA.id, frameB.id)
        if key in self.matches:
            return
        a, b = self._gt(frameA), self._gt(frameB)
        rng = np.random.default_rng([self.seq.seed, a, b])
        ca, cb = self.seq.poses_gt[a][:3, 3], self.seq.poses_gt[b][:3, 3]
        pts = np.zeros((0, 3))
        for _ in range(6):
            cand, nrm = _sample_surface(rng, max(4 * self.m, 2048))
            vis = (((ca - cand) * nrm).sum(1) > 0.02) & (((cb - cand) * nrm).sum(1) > 0.02)
            pts = np.concatenate([pts, cand[vis]])
            if pts.shape[0] >= self.m:
                break
        pts = pts[: self.m]
        n = pts.shape[0]
        pa = pts @ self._inv[a, :3, :3].T + self._inv[a, :3, 3] + rng.normal(scale=self.noise_m, size=(n, 3))
        pb = pts @ self._inv[b, :3, :3].T + self._inv[b, :3, 3] + rng.normal(scale=self.noise_m, size=(n, 3))
        n_out = int(round(self.outlier_frac * n))
        if n_out:
            idx = rng.choice(n, n_out, replace=False)
            dirs = rng.normal(size=(n_out, 3))
            dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
            pa[idx] += dirs * rng.uniform(0.02, 0.05, size=(n_out, 1))
        self.matches[key] = (pa.astype(np.float32), pb.astype(np.float32))
        if self.ransac is not None:
            self.ransac([(frameA, frameB)], self.matches)

    def procrustes_by_correspondence(self, frameA, frameB) -> np.ndarray:
        from .bundler import solve_rigid_transform_between_points
        ptA, ptB = self.matches.get((frameA.id, frameB.id), (np.zeros((0, 3), np.float32),) * 2)
        if len(ptA) < 5:
            return np.eye(4, dtype=np.float32)
        Ta, Tb = np.asarray(frameA.pose_in_model, np.float32), np.asarray(frameB.pose_in_model, np.float32)
        src = ptA @ Ta[:3, :3].T + Ta[:3, 3]
        dst = ptB @ Tb[:3, :3].T + Tb[:3, 3]
        pose = solve_rigid_transform_between_points(src, dst)
        res = np.linalg.norm(src @ pose[:3, :3].T + pose[:3, 3] - dst, axis=1)
        keep = res < self.inlier_dist
        if 5 <= keep.sum() < len(keep):
            pose = solve_rigid_transform_between_points(src[keep], dst[keep])
        return pose
