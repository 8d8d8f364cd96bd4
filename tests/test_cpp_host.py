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
    if not os.path.exists(_lib.HOST_DRIVER):
        _lib.build_host_cpp()
    return _lib.HOST_DRIVER


def test_cpp_keyframe_memory_matches_python(tmp_path):
    """checkAndAddKeyframe over a 40-frame orbit (4-6 degrees per frame) and selectKeyFramesForBA with a pool that does
    not fit max_BA_frames: the C++ and the Python restatement of Bundler.cpp:185-274 pick the same frames."""
    seq = S.SyntheticSequence(n_frames=40, seed=77, step_deg=(4.0, 6.0))
    poses = seq.poses_gt.astype(np.float32)
    for max_ba in (6, 15):
        inp, out = str(tmp_path / "kf_in.bin"), str(tmp_path / "kf_out.bin")
        with open(inp, "wb") as f:
            f.write(np.array([len(poses), max_ba], np.int32).tobytes()); f.write(poses.tobytes())
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
