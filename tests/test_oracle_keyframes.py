"""Keyframe memory: the two host implementations against the oracle's restatement of Bundler.cpp:185-274 and
Utils.cpp:42-47 (oracle/btba_oracle_keyframes.c -- PARITY UNPINNED: the reference's own functions need Eigen and yaml-cpp, which the
image lacks, and the reference has no vectors for them; the restatement is written from the reference text, statement by statement).
CPU only.  200 random pools: orbits with uneven steps, re-visited poses (exact ties, admitted with min_rot 0), near-ties, pools that
fit and pools that do not."""
import subprocess

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from bundletrack_amd import _lib
from bundletrack_amd.bundler import FrameRef, KeyframeMemory, rotation_geodesic_distance
from oracle import oracle as O


def driver():
    return _lib.build_host_cpp()          # (re)built when a source is newer


def make_pool(seed):
    """(poses [M, 4, 4] float32, min_rot_deg, max_BA_frames).  Every third pool re-visits earlier poses bit for bit and every
    sixth one admits them to the pool (min_rot 0): cum_dist then ties exactly and pool order has to decide (strict `<`, :257)."""
    rng = np.random.default_rng(seed)
    M = int(rng.integers(8, 48))
    step = rng.uniform(2.0, 16.0)
    axis0 = rng.normal(size=3); axis0 /= np.linalg.norm(axis0)
    R = Rotation.identity()
    poses = np.tile(np.eye(4, dtype=np.float32), (M, 1, 1))
    for k in range(M):
        if k:
            axis = axis0 + 0.3 * rng.normal(size=3); axis /= np.linalg.norm(axis)
            R = Rotation.from_rotvec(np.deg2rad(step * rng.uniform(0.3, 1.7)) * axis) * R
        poses[k, :3, :3] = R.as_matrix().astype(np.float32)
        poses[k, :3, 3] = rng.normal(size=3).astype(np.float32) * 0.05
    if seed % 3 == 0:
        for _ in range(int(rng.integers(2, 8))):
            a, b = rng.integers(1, M, size=2)
            poses[a] = poses[b]
    if seed % 7 == 0:                                         # near-ties: one ulp-sized nudge of a copy
        a, b = rng.integers(1, M, size=2)
        poses[a] = poses[b]; poses[a, 0, 0] = np.nextafter(poses[a, 0, 0], np.float32(0))
    min_rot = 0.0 if seed % 6 == 0 else float(rng.choice([5.0, 10.0, 10.0, 20.0]))
    max_ba = int(rng.choice([3, 4, 6, 10, 15]))
    return poses, min_rot, max_ba


def test_rotation_geodesic_distance_three_ways():
    """numpy host, and the oracle: same bits on random, identical and nearly identical rotations (the C++ host is held to the same
    values through the selections below: every cum_dist is a sum of these)."""
    rng = np.random.default_rng(5)
    for t in range(3000):
        a, b = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
        Ra = Rotation.random(random_state=int(rng.integers(1 << 31)))
        a[:3, :3] = Ra.as_matrix()
        b[:3, :3] = a[:3, :3] if t % 3 == 0 else (Rotation.from_rotvec(rng.normal(size=3) * 1e-3) * Ra).as_matrix() if t % 3 == 1 \
            else Rotation.random(random_state=int(rng.integers(1 << 31))).as_matrix()
        x, y = O.rotation_geodesic_distance(a, b), rotation_geodesic_distance(a[:3, :3], b[:3, :3])
        assert np.float32(x) == np.float32(y)
        assert abs(x - np.arccos(np.clip((np.trace(a[:3, :3].astype(np.float64) @ b[:3, :3].T.astype(np.float64)) - 1) / 2, -1, 1))) < 2e-3 * max(1e-3, 1 / max(x, 1e-3)) + 1e-6
    eye = np.eye(4, dtype=np.float32)
    assert O.rotation_geodesic_distance(eye, eye) == 0.0
    flip = np.diag([1, -1, -1, 1]).astype(np.float32)
    assert abs(O.rotation_geodesic_distance(eye, flip) - np.pi) < 1e-6                       # the clamp at -1 (Utils.cpp:45)


def test_keyframe_memory_against_the_oracle_on_200_pools(tmp_path):
    n_fit = n_greedy = n_tied_pools = n_addr_dependent = 0
    for seed in range(200):
        poses, min_rot, max_ba = make_pool(seed)
        M = len(poses)
        # the oracle: checkAndAddKeyframe over frames 0 .. M-2, selectKeyFramesForBA for frame M-1 (index order = id order = allocation order)
        added_o, pool_o = O.keyframe_pool(poses[:-1], min_rot_deg=min_rot)
        chosen_o = sorted(O.select_keyframes_for_ba(poses, M - 1, pool_o, max_ba).tolist())
        # numpy host
        mem = KeyframeMemory(min_rot_deg=min_rot, max_BA_frames=max_ba)
        frames = [FrameRef(id=k, pose_in_model=poses[k], n_keypts=100) for k in range(M)]
        added_py = [mem.check_and_add_keyframe(fr) for fr in frames[:-1]]
        chosen_py = [f.id for f in mem.select_keyframes_for_ba(frames[-1])]
        assert added_py == added_o.tolist(), seed
        assert chosen_py == chosen_o, (seed, chosen_py, chosen_o)
        # C++ host
        inp, out = str(tmp_path / "kf_in.bin"), str(tmp_path / "kf_out.bin")
        with open(inp, "wb") as f:
            f.write(np.array([M, max_ba], np.int32).tobytes()); f.write(np.float32(min_rot).tobytes()); f.write(poses.tobytes())
        subprocess.run([driver(), "keyframes", inp, out], check=True, timeout=60)
        res = np.fromfile(out, np.int32)
        split = int(np.nonzero(res == -1)[0][0])
        assert res[:split].astype(bool).tolist() == added_o.tolist(), seed
        assert res[split + 1:].tolist() == chosen_o, (seed, res[split + 1:].tolist(), chosen_o)
        # what was exercised
        assert len(chosen_o) == min(max_ba, len(pool_o) + 1) and (M - 1) in chosen_o
        if len(pool_o) + 1 <= max_ba:
            n_fit += 1
            assert chosen_o == sorted(pool_o.tolist() + [M - 1])                         # :227-235
        else:
            n_greedy += 1
            assert int(pool_o[0]) in chosen_o                                            # :237
            n_tied_pools += len({poses[k].tobytes() for k in pool_o}) < len(pool_o)
            # the reference's address order only enters through the summation order of cum_dist: a different order of the
            # frames in memory may move a choice between near-equal candidates, and nothing else
            rng = np.random.default_rng(seed)
            for _ in range(3):
                alt = sorted(O.select_keyframes_for_ba(poses, M - 1, pool_o, max_ba, addr_rank=rng.permutation(M)).tolist())
                n_addr_dependent += alt != chosen_o
    assert n_fit >= 20 and n_greedy >= 100 and n_tied_pools >= 10, (n_fit, n_greedy, n_tied_pools)
    assert n_addr_dependent <= 0.05 * 3 * n_greedy, n_addr_dependent


def test_check_and_add_keyframe_gates():
    """frame 0 always; status / keypoint gates (Bundler.cpp:187-202) in the oracle as in the numpy host."""
    poses = np.tile(np.eye(4, dtype=np.float32), (4, 1, 1))
    for k in range(4):
        poses[k, :3, :3] = Rotation.from_euler("y", 15.0 * k, degrees=True).as_matrix()
    added, pool = O.keyframe_pool(poses, status_other=[0, 1, 0, 1], n_keypts=[0, 5, 500, 500], min_feat_num=10)
    assert added.tolist() == [True, False, False, True] and pool.tolist() == [0, 3]
    mem = KeyframeMemory(min_feat_num=10)
    got = [mem.check_and_add_keyframe(FrameRef(id=k, pose_in_model=poses[k], n_keypts=n, status=s))
           for k, (n, s) in enumerate(zip([0, 5, 500, 500], ["FAIL", "OTHER", "NO_BA", "OTHER"]))]
    assert got == added.tolist()
