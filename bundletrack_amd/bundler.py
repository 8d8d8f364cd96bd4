"""Host-side caller logic either side of the optimiser boundary (A0 and 8(f) rank 2).

Restates, in numpy, the parts of src/Bundler.cpp a drop-in user needs around the hot path:
  * `marshal_window`            Bundler::optimizeGPU's marshalling      (Bundler.cpp:286-347)
  * `check_and_add_keyframe`    Bundler::checkAndAddKeyframe            (Bundler.cpp:185-219)
  * `select_keyframes_for_ba`   Bundler::selectKeyFramesForBA           (Bundler.cpp:222-274)
  * `rotation_geodesic_distance` Utils::rotationGeodesicDistance        (Utils.cpp:42-47)
Feature detection, matching and RANSAC (FeatureManager.*) stay out of scope: `matches` arrive
as arrays of 3-D point pairs, which is exactly what `Correspondence._ptA_cam/_ptB_cam` carry.
"""
from __future__ import annotations

import ctypes
import ctypes.util
from dataclasses import dataclass, field

import numpy as np

from ._lib import ENTRYJ_DTYPE

# std::acos(float) of the reference is the C library's acosf; numpy's float32 arccos is its own vector routine and differs from it in the
# last bit for about one argument in five -- enough to flip a greedy choice between near-equal candidates
_acosf = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6").acosf
_acosf.argtypes, _acosf.restype = [ctypes.c_float], ctypes.c_float


def rotation_geodesic_distance(R1: np.ndarray, R2: np.ndarray) -> float:
    """acos(clamp((tr(R1 R2^T) - 1) / 2)) in fp32 like the Eigen::Matrix3f original (Utils.cpp:42-47): diagonal entry i of
    R1 R2^T is the dot product of the two i-th rows, each summed left to right in float and then the three of them (no BLAS
    call whose summation order or FMA use depends on the host; oracle/btba_oracle_keyframes.c sums the same way)."""
    R1 = np.asarray(R1, np.float32)
    R2 = np.asarray(R2, np.float32)
    prod = R1 * R2                                            # float32 products, each rounded once
    dots = (prod[:, 0] + prod[:, 1]) + prod[:, 2]
    trace = (dots[0] + dots[1]) + dots[2]
    tmp = np.float32((trace - np.float32(1.0)) / 2.0)
    tmp = max(min(np.float32(1.0), tmp), np.float32(-1.0))
    return float(_acosf(float(tmp)))


@dataclass
class FrameRef:
    """The fields of `Frame` (src/Frame.h:45-96) the BA caller touches."""
    id: int
    pose_in_model: np.ndarray            # [4,4] camera -> model
    status: str = "OTHER"                # Frame::Status {FAIL, NO_BA, OTHER}, Frame.h:48-53
    n_keypts: int = 0
    roi: tuple = (0.0, 1e9, 0.0, 1e9)    # (umin, umax, vmin, vmax) of the segmentation mask, Frame.h:81
    depth_gpu: object = None
    normal_gpu: object = None
    color_gpu: object = None

    def __hash__(self):
        return hash(self.id)


@dataclass
class KeyframeMemory:
    """The "memory" of the memory-augmented pose graph: the keyframe pool and the BA subset."""
    min_rot_deg: float = 10.0            # keyframe.min_rot      (config_ycbineoat.yml:36)
    min_feat_num: int = 0                # keyframe.min_feat_num (:35)
    max_BA_frames: int = 15              # bundle.max_BA_frames  (:27)
    keyframes: list = field(default_factory=list)

    def check_and_add_keyframe(self, frame: FrameRef) -> bool:
        """Bundler.cpp:185-219: frame 0 always; otherwise only status OTHER, enough keypoints, and a
        rotation >= min_rot degrees away from EVERY existing keyframe."""
        if frame.id == 0:
            self.keyframes.append(frame)
            return True
        if frame.status != "OTHER":
            return False
        if frame.n_keypts < self.min_feat_num:
            return False
        for kf in self.keyframes:
            rot_diff = rotation_geodesic_distance(frame.pose_in_model[:3, :3], kf.pose_in_model[:3, :3])
            rot_diff = np.float32(np.float64(np.float32(rot_diff) * np.float32(180.0)) / np.pi)     # product in float, division in double (:208)
            if rot_diff < self.min_rot_deg:
                return False
        self.keyframes.append(frame)
        return True

    def select_keyframes_for_ba(self, newframe: FrameRef) -> list:
        """Bundler.cpp:222-274 ("greedy_rot"): the new frame plus, if the pool does not fit, keyframe 0
        and then repeatedly the keyframe with the SMALLEST summed geodesic rotation distance to the
        already chosen set.  (The reference keeps the chosen set in a std::set of shared_ptr, i.e. in
        pointer order; the result is sorted by frame id afterwards -- Bundler.cpp:286 -- so only the
        summation order of cum_dist depends on it; here the set is iterated in frame-id order: what an
        allocator that hands out ascending addresses gives the reference.)"""
        chosen = [newframe]
        if len(self.keyframes) + len(chosen) <= self.max_BA_frames:
            for kf in self.keyframes:
                if kf not in chosen:
                    chosen.append(kf)
            return sorted(chosen, key=lambda f: f.id)
        if self.keyframes[0] not in chosen:
            chosen.append(self.keyframes[0])
        while len(chosen) < self.max_BA_frames:
            best, best_kf = np.finfo(np.float32).max, None
            for kf in self.keyframes:
                if kf in chosen:
                    continue
                cum = np.float32(0)
                for f in sorted(chosen, key=lambda f: f.id):
                    cum = np.float32(cum + np.float32(rotation_geodesic_distance(kf.pose_in_model[:3, :3], f.pose_in_model[:3, :3])))
                if cum < best:
                    best, best_kf = cum, kf
            if best_kf is None:
                break
            chosen.append(best_kf)
        return sorted(chosen, key=lambda f: f.id)


@dataclass
class Window:
    """What optimizeGPU hands to OptimizerGpu::optimizeFrames."""
    frames: list
    corr: np.ndarray                  # ENTRYJ_DTYPE, pair-major
    n_match_per_pair: np.ndarray
    n_edges_newframe: int
    run_ba: bool                      # False <=> Frame::NO_BA (Bundler.cpp:343-347)


def marshal_window(local_frames, matches, newframe, min_fm_edges_newframe: int = 5) -> Window:
    """Bundler::optimizeGPU up to the optimiser call (Bundler.cpp:286-347).

    local_frames: FrameRef list (any order; sorted by id here like :286, so index 0 = oldest = fixed).
    matches: dict {(id_A, id_B): (ptA_cam [m,3], ptB_cam [m,3])} keyed by (later frame id, earlier
    frame id) like `_fm->_matches[{frameA, frameB}]` with frameA = local_frames[j], frameB =
    local_frames[i], i<j.  EntryJ{imgIdx_i=i, imgIdx_j=j, pos_i=ptB_cam, pos_j=ptA_cam} (:311-316)."""
    frames = sorted(local_frames, key=lambda f: f.id)
    blocks, counts, n_edges = [], [], 0
    for i in range(len(frames)):
        for j in range(i + 1, len(frames)):
            fa, fb = frames[j], frames[i]
            ptA, ptB = matches.get((fa.id, fb.id), (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)))
            m = len(ptA)
            blk = np.zeros(m, ENTRYJ_DTYPE)
            blk["imgIdx_i"], blk["imgIdx_j"] = i, j
            blk["pos_i"], blk["pos_j"] = np.asarray(ptB, np.float32).reshape(m, 3), np.asarray(ptA, np.float32).reshape(m, 3)
            blocks.append(blk)
            counts.append(m)
            if fa.id == newframe.id or fb.id == newframe.id:
                n_edges += m
    corr = np.concatenate(blocks) if blocks else np.zeros(0, ENTRYJ_DTYPE)
    run = n_edges > min_fm_edges_newframe
    if not run:
        newframe.status = "NO_BA"
    return Window(frames=frames, corr=corr, n_match_per_pair=np.asarray(counts, np.int32), n_edges_newframe=n_edges, run_ba=run)


# ---------------------------------------------------------------------------------------------------------
# The rest of Bundler's BA-facing slice: Kabsch initialisation, the per-frame driver, the pose-file format.
# Feature detection / matching / RANSAC (LF-Net, SiftGPU, cuda_ransac) stay behind `feature_manager`.
# ---------------------------------------------------------------------------------------------------------

def solve_rigid_transform_between_points(points1: np.ndarray, points2: np.ndarray) -> np.ndarray:
    """Utils::solveRigidTransformBetweenPoints (src/Utils.cpp:180-214): Kabsch, points1 -> points2, fp32.
    Identity when V U^T is not orthonormal or the result is not finite; a reflection flips V's last column."""
    p1 = np.asarray(points1, np.float32).reshape(-1, 3)
    p2 = np.asarray(points2, np.float32).reshape(-1, 3)
    pose = np.eye(4, dtype=np.float32)
    if p1.shape[0] < 3 or p1.shape != p2.shape:
        return pose
    m1, m2 = p1.mean(0, dtype=np.float32), p2.mean(0, dtype=np.float32)
    S = (p1 - m1).T @ (p2 - m2)
    if not np.isfinite(S).all():            # Eigen's JacobiSVD would return NaNs and the isApprox test below would fail
        return pose
    U, _, Vt = np.linalg.svd(S.astype(np.float32))
    V = Vt.T
    R = V @ U.T
    if not np.allclose(R.T @ R, np.eye(3, dtype=np.float32), rtol=1e-5, atol=1e-5):       # Eigen isApprox, float precision
        return pose
    if np.linalg.det(R) < 0:
        V = V.copy()
        V[:, 2] = -V[:, 2]
        R = V @ U.T
    pose[:3, :3] = R
    pose[:3, 3] = m2 - R @ m1
    if not np.isfinite(pose).all():
        return np.eye(4, dtype=np.float32)
    return pose


def format_pose_txt(ob_in_cam: np.ndarray) -> str:
    """`ff << std::setprecision(10) << ob_in_cam << std::endl` (Bundler.cpp:372-377) with Eigen's default
    IOFormat: coefficients in %.10g, every column padded to the widest coefficient, single-space separator."""
    M = np.asarray(ob_in_cam, np.float32).reshape(4, 4)
    cells = [[format(float(v), ".10g") for v in row] for row in M]
    width = max(len(c) for row in cells for c in row)
    return "\n".join(" ".join(c.rjust(width) for c in row) for row in cells) + "\n"


def save_pose_txt(path: str, pose_in_model: np.ndarray) -> None:
    """saveNewframeResult's pose file: poses/<id_str>.txt holds ob_in_cam = inverse(camera -> model)."""
    ob_in_cam = np.linalg.inv(np.asarray(pose_in_model, np.float32)).astype(np.float32)
    with open(path, "w") as f:
        f.write(format_pose_txt(ob_in_cam))


def load_pose_txt(path: str) -> np.ndarray:
    """What scripts/eval_ycbineoat.py:124-144 does with the file (np.loadtxt -> 4x4)."""
    return np.loadtxt(path).reshape(4, 4)


class Bundler:
    """Bundler::processNewFrame (src/Bundler.cpp:52-183) from the point where a frame has a mask and features:
    pose initialisation from the previous frame, sliding window, keyframe subset, bundle adjustment through
    an injected `optimizer` (anything with OptimizerGpu.optimizeFrames' signature) and keyframe insertion.

    feature_manager must offer (the slice of SiftManager the caller uses):
        find_corres(frameA, frameB)        -> None; fills matches[(frameA.id, frameB.id)] = (ptA_cam, ptB_cam), A newer
        matches                            -> dict as above
        procrustes_by_correspondence(frameA, frameB) -> 4x4 model-frame offset (FeatureManager.cpp:523-556)
        forget_frame(frame)                -> None
    """

    def __init__(self, optimizer, feature_manager, K, H, W, *, window_size=2, max_BA_frames=15, min_rot_deg=10.0,
                 min_feat_num=0, min_fm_edges_newframe=5, pose_dir=None, persistent_frame_cache=False):
        self.opt, self.fm = optimizer, feature_manager
        self.K, self.H, self.W = np.asarray(K, np.float32), int(H), int(W)
        self.window_size = int(window_size)                        # bundle.window_size (config_ycbineoat.yml:26 ships 2)
        self.min_fm_edges_newframe = int(min_fm_edges_newframe)    # bundle.min_fm_edges_newframe
        self.memory = KeyframeMemory(min_rot_deg=min_rot_deg, min_feat_num=min_feat_num, max_BA_frames=max_BA_frames)
        self.frames: list = []                                     # _frames (deque)
        self.local_frames: list = []
        self.newframe = None
        self.need_reinit = False
        self.pose_dir = pose_dir
        self.persistent_frame_cache = bool(persistent_frame_cache)     # hand Frame ids to the optimiser as cache keys
        if self.persistent_frame_cache and getattr(optimizer, "workspace", None) is not None:
            from .optimizer import frame_cache_clear
            frame_cache_clear(optimizer.workspace)                     # frame ids restart at 0 with every tracking session
        self.n_ba_calls = 0
        self.last_window = None

    @property
    def keyframes(self):
        return self.memory.keyframes

    def process_new_frame(self, frame: FrameRef) -> None:
        self.newframe = frame
        last = self.frames[-1] if self.frames else None
        if last is not None:
            frame.id = last.id + 1
            frame.pose_in_model = np.array(last.pose_in_model, np.float32)             # :78-79
        if frame.roi[1] - frame.roi[0] < 10 or frame.roi[3] - frame.roi[2] < 10:        # :88-93: empty cloud -> FAIL and a plain return
            frame.status = "FAIL"
            return
        if frame.status == "FAIL":                                                     # :96-101
            self.fm.forget_frame(frame)
            self.need_reinit = True
            return
        if last is not None:
            self.fm.find_corres(frame, last)                                           # :122
            if frame.status == "FAIL":
                self.need_reinit = True
                self.fm.forget_frame(frame)
                return
            offset = self.fm.procrustes_by_correspondence(frame, last)                 # :134-136
            frame.pose_in_model = (np.asarray(offset, np.float32) @ frame.pose_in_model).astype(np.float32)
        if len(self.frames) >= self.window_size + 3:                                   # :150-158
            if self.frames[0] not in self.keyframes:
                self.fm.forget_frame(self.frames[0])
            self.frames.pop(0)
        self.frames.append(frame)
        if frame.id == 0:
            self.memory.check_and_add_keyframe(frame)
            return
        self.local_frames = self.memory.select_keyframes_for_ba(frame)                 # :168
        self.optimize_gpu()                                                            # :169
        if frame.status == "FAIL":
            self.fm.forget_frame(frame)
            self.frames.pop()
            self._evict_cached(frame)                  # its id is handed out again to the next frame (last.id + 1)
            self.need_reinit = True
            return
        self.memory.check_and_add_keyframe(frame)
        if self.pose_dir is not None:
            self.save_newframe_result()

    def _evict_cached(self, frame) -> None:
        """A frame dropped after BA has cached it must not leave its (z, n) cache behind under an id the next frame reuses."""
        if self.persistent_frame_cache and getattr(self.opt, "workspace", None) is not None:
            from .optimizer import frame_cache_evict
            frame_cache_evict(self.opt.workspace, frame.id)

    def optimize_gpu(self) -> None:
        """Bundler::optimizeGPU (:279-359): match every pair of the window, marshal, gate, optimise, write back."""
        frames = sorted(self.local_frames, key=lambda f: f.id)
        for i in range(len(frames)):
            for j in range(i + 1, len(frames)):
                self.fm.find_corres(frames[j], frames[i])
        win = marshal_window(frames, self.fm.matches, self.newframe, self.min_fm_edges_newframe)
        self.last_window = win
        if not win.run_ba:
            return
        poses = np.stack([np.asarray(f.pose_in_model, np.float32) for f in win.frames])
        extra = {"frame_keys": [f.id for f in win.frames]} if self.persistent_frame_cache else {}
        self.opt.optimizeFrames(win.corr, win.n_match_per_pair, len(win.frames), self.H, self.W,
                                [f.depth_gpu for f in win.frames], [f.color_gpu for f in win.frames],
                                [f.normal_gpu for f in win.frames], poses, self.K, **extra)
        self.n_ba_calls += 1
        for f, T in zip(win.frames, poses):
            f.pose_in_model = np.array(T, np.float32)

    def save_newframe_result(self) -> None:
        import os
        os.makedirs(self.pose_dir, exist_ok=True)
        name = getattr(self.newframe, "id_str", None) or "%04d" % self.newframe.id
        save_pose_txt(os.path.join(self.pose_dir, name + ".txt"), self.newframe.pose_in_model)
