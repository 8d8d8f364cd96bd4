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

from dataclasses import dataclass, field

import numpy as np

from ._lib import ENTRYJ_DTYPE


def rotation_geodesic_distance(R1: np.ndarray, R2: np.ndarray) -> float:
    """acos(clamp((tr(R1 R2^T) - 1) / 2)) in fp32 like the Eigen::Matrix3f original (Utils.cpp:42-47)."""
    R1 = np.asarray(R1, np.float32)
    R2 = np.asarray(R2, np.float32)
    tmp = np.float32((np.trace(R1 @ R2.T) - np.float32(1.0)) / 2.0)
    tmp = max(min(np.float32(1.0), tmp), np.float32(-1.0))
    return float(np.arccos(np.float32(tmp)))


@dataclass
class FrameRef:
    """The fields of `Frame` (src/Frame.h:45-96) the BA caller touches."""
    id: int
    pose_in_model: np.ndarray            # [4,4] camera -> model
    status: str = "OTHER"                # Frame::Status {FAIL, NO_BA, OTHER}, Frame.h:48-53
    n_keypts: int = 0
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
            rot_diff = np.float32(rot_diff) * np.float32(180.0) / np.float32(np.pi)
            if rot_diff < self.min_rot_deg:
                return False
        self.keyframes.append(frame)
        return True

    def select_keyframes_for_ba(self, newframe: FrameRef) -> list:
        """Bundler.cpp:222-274 ("greedy_rot"): the new frame plus, if the pool does not fit, keyframe 0
        and then repeatedly the keyframe with the SMALLEST summed geodesic rotation distance to the
        already chosen set.  (The reference keeps the chosen set in a std::set of shared_ptr, i.e. in
        pointer order; the result is sorted by frame id afterwards -- Bundler.cpp:286 -- so only the
        summation order of cum_dist depends on it; here the set is iterated in insertion order.)"""
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
                for f in chosen:
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
