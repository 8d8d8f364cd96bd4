"""The problem-dump format (bundletrack_amd/problem_io.py; SURVEY.md 8(d) "problem dumps are the interchange format", 8(f) row 4):
written and read by Python, read and re-written by the C++ host layer (btba::loadProblem / saveProblem) byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from bundletrack_amd import _lib, problem_io as IO, synthetic as S


@pytest.fixture(scope="module")
def window():
    Ks = S.NOCS_K.copy(); Ks[:2] *= 0.2
    return S.make_problem(4, 30, seed=3, background=False, H=96, W=128, K=Ks)


def test_round_trip_in_python(window, tmp_path):
    p = str(tmp_path / "w.btba")
    IO.save_problem(p, window)
    N, H, W, C = 4, 96, 128, len(window.corr)
    assert os.path.getsize(p) == 32 + 36 + 32 * C + 4 * 6 + 64 * N + 128 * N + 4 * N * H * W + 16 * N * H * W      # the documented layout
    q = IO.load_problem(p)
    assert (q.n_frames, q.H, q.W, q.image_downscale) == (N, H, W, 4.0)
    assert np.array_equal(q.K, window.K.astype(np.float32)) and np.array_equal(q.corr, window.corr)
    assert np.array_equal(q.n_match_per_pair, window.n_match_per_pair) and np.array_equal(q.poses_init, window.poses_init)
    assert np.array_equal(q.poses_gt, window.poses_gt) and np.array_equal(q.depth, window.depth) and np.array_equal(q.normals, window.normals)
    p2 = str(tmp_path / "w2.btba")
    IO.save_problem(p2, q)                                       # a loaded dump saves to the same bytes
    assert open(p, "rb").read() == open(p2, "rb").read()
    q.poses_gt = None                                           # ground truth is optional (a real tracker has none)
    IO.save_problem(p2, q)
    r = IO.load_problem(p2)
    assert r.poses_gt is None and np.array_equal(r.depth, window.depth)


def test_malformed_dumps_are_refused(window, tmp_path):
    p = str(tmp_path / "w.btba")
    IO.save_problem(p, window)
    raw = open(p, "rb").read()
    for name, data in (("magic", b"BTBAPRB0" + raw[8:]), ("short", raw[:-5]), ("long", raw + b"\0"), ("tiny", raw[:40])):
        bad = str(tmp_path / f"{name}.btba")
        open(bad, "wb").write(data)
        with pytest.raises(ValueError):
            IO.load_problem(bad)
    with pytest.raises(ValueError):                             # cache-resolution problems carry no frames to dump
        IO.save_problem(p, S.make_problem(3, 10, seed=1, full_res=False))


def test_cpp_host_layer_reads_and_writes_the_same_bytes(window, tmp_path):
    if not os.path.exists(_lib.HOST_DRIVER):
        _lib.build_host_cpp()
    p, copy = str(tmp_path / "w.btba"), str(tmp_path / "copy.btba")
    IO.save_problem(p, window)
    r = subprocess.run([_lib.HOST_DRIVER, "problem", p, copy], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert open(p, "rb").read() == open(copy, "rb").read()
    valid = int((window.depth >= 0.1).sum())
    assert f"4 frames 128x96, {len(window.corr)} correspondences (longest pair segment 30), {valid} valid depth pixels, ground truth yes" in r.stdout
    bad = str(tmp_path / "bad.btba")
    open(bad, "wb").write(open(p, "rb").read()[:-7])
    r = subprocess.run([_lib.HOST_DRIVER, "problem", bad, copy], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "truncated" in r.stderr        # btba::Error, not a crash
