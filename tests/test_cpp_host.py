"""The C++ host layer above the C ABI (bundletrack_amd/cpp/btba_host.*, the reference's host code is C++ too): its
OptimizerGpu / marshalWindow / KeyframeMemory are driven through tests/cpp/host_driver and compared with the Python
mirror and the oracle."""
import os
import subprocess

import numpy as np
import pytest

from bundletrack_amd import _lib, synthetic as S
from bundletrack_amd.bundler import FrameRef, KeyframeMemory


def driver():
    return _lib.build_host_cpp()          # (re)built when a source is newer


def test_cpp_keyframe_memory_matches_python(tmp_path):
    """checkAndAddKeyframe over a 40-frame orbit (4-6 degrees per frame) and selectKeyFramesForBA with a pool that does
    not fit max_BA_frames: the C++ and the Python restatement of Bundler.cpp:185-274 pick the same frames."""
    seq = S.SyntheticSequence(n_frames=40, seed=77, step_deg=(4.0, 6.0))
    poses = seq.poses_gt.astype(np.float32)
    for max_ba in (6, 15):
        inp, out = str(tmp_path / "kf_in.bin"), str(tmp_path / "kf_out.bin")
        with open(inp, "wb") as f:
            f.write(np.array([len(poses), max_ba], np.int32).tobytes()); f.write(np.float32(10.0).tobytes()); f.write(poses.tobytes())
        subprocess.run([driver(), "keyframes", inp, out], check=True, timeout=60)
        res = np.fromfile(out, np.int32)
        split = int(np.nonzero(res == -1)[0][0])
        added_cpp, chosen_cpp = res[:split].astype(bool), res[split + 1:].tolist()
        mem = KeyframeMemory(max_BA_frames=max_ba)
        frames = [FrameRef(id=k, pose_in_model=poses[k], n_keypts=100) for k in range(len(poses))]
        added_py = [mem.check_and_add_keyframe(fr) for fr in frames[:-1]]
        chosen_py = [f.id for f in mem.select_keyframes_for_ba(frames[-1])]
        assert added_cpp.tolist() == added_py and 5 < sum(added_py) < 39
        assert chosen_cpp == chosen_py and len(chosen_py) == min(max_ba, sum(added_py) + 1)


@pytest.mark.gpu
def test_cpp_optimizer_gpu_matches_python_and_oracle(oracle, tmp_path):
    """A window dumped to disk, loaded by the C++ driver (hipMalloc'd frames, frames handed over in shuffled order with
    non-contiguous ids), marshalled and optimised through btba::OptimizerGpu, stateless and with the persistent frame
    cache: bit-identical to the Python mirror, within the bar of the oracle."""
    import torch
    from bundletrack_amd.optimizer import OptimizerGpu
    pb = S.make_problem(5, 200, seed=81, background=False)
    inp, out = str(tmp_path / "pb.bin"), str(tmp_path / "poses.bin")
    with open(inp, "wb") as f:
        f.write(np.array([pb.n_frames, pb.H, pb.W, len(pb.corr)], np.int32).tobytes())
        f.write(pb.K.astype(np.float32).tobytes()); f.write(pb.corr.tobytes()); f.write(pb.poses_init.astype(np.float32).tobytes())
        for k in range(pb.n_frames):
            f.write(pb.depth[k].tobytes()); f.write(pb.normals[k].tobytes())
    r = subprocess.run([driver(), "ba", inp, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert "cached in this call 5" in lines[0] and "cached in this call 5" in lines[1] and "cached in this call 0" in lines[2], lines
    res = np.fromfile(out, np.float32).reshape(3, pb.n_frames, 4, 4)
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(pb.depth[k]).to(dev) for k in range(pb.n_frames)]
    n = [torch.from_numpy(pb.normals[k]).to(dev) for k in range(pb.n_frames)]
    poses = pb.poses_init.copy()
    OptimizerGpu().optimizeFrames(pb.corr, pb.n_match_per_pair, pb.n_frames, pb.H, pb.W, d, None, n, poses, pb.K)
    assert np.array_equal(res[0], poses)
    caches = [oracle.build_cache(pb.depth[k], pb.normals[k], pb.K) for k in range(pb.n_frames)]
    ref = oracle.solve(np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches]), caches[0]["intr"], pb.corr, pb.poses_init)
    assert max(max(S.pose_error(res[0][k], ref.poses[k])) for k in range(pb.n_frames)) < 1e-4


def test_cpp_kabsch_matches_python(tmp_path):
    """Utils::solveRigidTransformBetweenPoints (Utils.cpp:180-214) in the C++ host layer against the Python restatement:
    noisy rigid motions, coplanar and minimal (3-point) sets, a mirrored set (the det < 0 branch), and the identity
    fall-backs (fewer than 3 points, non-finite input)."""
    from bundletrack_amd.bundler import solve_rigid_transform_between_points as kabsch_py
    rng = np.random.default_rng(11)
    sets = []
    for trial in range(40):
        n = int(rng.choice([3, 4, 10, 200]))
        a = rng.normal(scale=0.1, size=(n, 3)).astype(np.float32) + np.float32([0.0, 0.0, 0.7])
        if trial % 5 == 1:
            a[:, 2] = 0.7 + 0.3 * a[:, 0]                       # coplanar: rank-2 covariance
        T = S.se3_exp(rng.normal(scale=0.5, size=3), rng.normal(scale=0.1, size=3)) if hasattr(S, "se3_exp") else None
        if T is None:
            w = rng.normal(scale=0.5, size=3); th = np.linalg.norm(w); k = w / th
            Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
            t = rng.normal(scale=0.1, size=3)
        else:
            R, t = T[:3, :3], T[:3, 3]
        b = (a @ R.T + t + rng.normal(scale=0.0005, size=a.shape)).astype(np.float32)
        if trial % 7 == 3:
            b[:, 0] = -b[:, 0]                                  # mirrored target: V's last column is flipped
        sets.append((a, b))
    sets.append((sets[0][0][:2], sets[0][1][:2]))               # two points: identity
    bad = sets[1][0].copy(); bad[0, 0] = np.nan
    sets.append((bad, sets[1][1]))                              # non-finite: identity
    inp, out = str(tmp_path / "kabsch_in.bin"), str(tmp_path / "kabsch_out.bin")
    with open(inp, "wb") as f:
        f.write(np.array([len(sets)], np.int32).tobytes())
        for a, b in sets:
            f.write(np.array([len(a)], np.int32).tobytes()); f.write(np.ascontiguousarray(a, np.float32).tobytes()); f.write(np.ascontiguousarray(b, np.float32).tobytes())
    subprocess.run([driver(), "kabsch", inp, out], check=True, timeout=60)
    got = np.fromfile(out, np.float32).reshape(len(sets), 4, 4)
    worst = 0.0
    for k, (a, b) in enumerate(sets):
        ref = kabsch_py(a, b)
        assert np.isfinite(got[k]).all() and abs(np.linalg.det(got[k][:3, :3].astype(np.float64)) - 1.0) < 1e-4
        if len(a) > 3 or k >= len(sets) - 2:                    # 3 points are coplanar AND the fit is exact either way; compare the fit below
            worst = max(worst, float(np.abs(got[k] - ref).max()))
        if k >= len(sets) - 2:
            continue                                            # the identity fall-backs: nothing was fitted
        res_cpp = np.linalg.norm(a @ got[k][:3, :3].T + got[k][:3, 3] - b, axis=1).mean()
        res_py = np.linalg.norm(a @ ref[:3, :3].T + ref[:3, 3] - b, axis=1).mean()
        assert res_cpp <= res_py + 2e-6, (k, res_cpp, res_py)
    assert np.array_equal(got[-1], np.eye(4, dtype=np.float32)) and np.array_equal(got[-2], np.eye(4, dtype=np.float32))
    assert worst < 2e-5, worst


def test_cpp_pose_txt_matches_python(tmp_path):
    """The pose file body (Bundler.cpp:372-377: setprecision(10), Eigen's default IOFormat) from the C++ host layer and from
    bundler.format_pose_txt, character for character."""
    from bundletrack_amd.bundler import format_pose_txt
    rng = np.random.default_rng(3)
    poses = [np.eye(4, dtype=np.float32)]
    for _ in range(20):
        M = rng.normal(size=(4, 4)).astype(np.float32) * np.float32(10.0 ** rng.integers(-6, 4))
        M[3] = [0, 0, 0, 1]
        poses.append(M)
    poses.append(np.linalg.inv(S.SyntheticSequence(n_frames=3, seed=1).poses_gt[2]).astype(np.float32))
    inp, out = str(tmp_path / "poses.bin"), str(tmp_path / "poses.txt")
    with open(inp, "wb") as f:
        f.write(np.array([len(poses)], np.int32).tobytes()); f.write(np.stack(poses).astype(np.float32).tobytes())
    subprocess.run([driver(), "posetxt", inp, out], check=True, timeout=60)
    assert open(out).read() == "".join(format_pose_txt(P) for P in poses)
