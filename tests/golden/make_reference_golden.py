"""Generates tests/golden/ref_*.npz: golden vectors produced by THE REFERENCE ITSELF.

The reference's solver (solveBundlingStub and all its kernels, src/cuda/Solver/SolverBundling.cu), its frame-cache kernels
and its SE(3) helpers are compiled for the CPU from the sources under /root/reference and executed through the sequential
launch emulator (oracle/Makefile target `ref`, oracle/ref_shim/, oracle/reference.py).  This script feeds them seeded
synthetic windows and records inputs and outputs: the frame cache of every frame, the poses after 1..7 Gauss-Newton
iterations, a table of Exp / Log values across all branch thresholds.  The files travel with the repository, so the
oracle (CPU tests) and the HIP path (GPU tests) are checked against the reference's own numbers even where neither the
reference checkout nor oracle/_ref exists.  Needs /root/reference (or prebuilt oracle/_ref/*.so):

    python tests/golden/make_reference_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bundletrack_amd import synthetic as S  # noqa: E402
from oracle import reference as R  # noqa: E402

CASES = {
    # name: (n_frames, corr/pair, seed, background, weight_dense, (H, W))          cache = frame / 4
    "ref_k4_masked": (4, 80, 201, False, 1.0, (192, 256)),
    "ref_k5_masked": (5, 60, 202, False, 1.0, (192, 256)),
    "ref_k6_features_only": (6, 120, 203, False, 0.0, (192, 256)),
    "ref_k3_full": (3, 100, 204, True, 1.0, (96, 128)),
}


def kinv4(K):
    K4 = np.eye(4, dtype=np.float32); K4[:3, :3] = K
    return R.mat4_inverse(K4)


def make(name):
    n, m, seed, bg, wd, (H, W) = CASES[name]
    K = S.NOCS_K * np.array([[W / 640.0], [H / 480.0], [1.0]])
    pb = S.make_problem(n, m, seed, background=bg, H=H, W=W, downscale=4, K=K)
    frames = [R.store_frame(pb.depth[k], pb.normals[k], kinv4(pb.K), 4.0) for k in range(n)]
    campos = np.stack([f[0] for f in frames]); normals = np.stack([f[1] for f in frames]); n_valid = np.array([f[3] for f in frames], np.int32)
    Hd, Wd = campos.shape[1:3]
    intr = np.array([pb.K[0, 0] * (Wd / W), pb.K[1, 1] * (Hd / H), pb.K[0, 2] * ((Wd - 1) / (W - 1)), pb.K[1, 2] * ((Hd - 1) / (H - 1))], np.float32)   # CUDACache.cpp:21-24
    poses = np.stack([R.solve(campos, normals, intr, pb.corr, pb.poses_init, n_gn=it, weight_dense=wd)[0] for it in range(1, 8)])
    return dict(depth=pb.depth, normals_full=pb.normals, K=pb.K, campos=campos, normals=normals, n_valid=n_valid, intr=intr,
                corr=pb.corr.view(np.uint8).reshape(-1, 32), poses_init=pb.poses_init, weight_dense=np.float32(wd), poses_after=poses)


def se3_table():
    rng = np.random.default_rng(7)
    rots = [np.zeros(3), [1e-5, 0, 0], [9e-5, 3e-5, 0], [3e-4, 0, 0], [1e-3, -1e-3, 2e-4], [0.03, 0.01, -0.02], [0.7, -0.4, 0.2], [2.0, 1.5, -1.0],
            [3.1, 0.2, 0.1], [0, 3.14159, 0], [3.1415, 0, 0]] + list(rng.uniform(-np.pi, np.pi, size=(53, 3)))
    rots = np.array(rots, np.float32)
    trans = rng.uniform(-1, 1, size=rots.shape).astype(np.float32)
    M = np.stack([R.pose_to_matrix(r, t) for r, t in zip(rots, trans)])
    back = np.stack([np.concatenate(R.matrix_to_pose(m)) for m in M])
    inv = np.stack([R.mat4_inverse(m) for m in M])
    return dict(rot=rots, trans=trans, matrix=M, log=back, inverse=inv)


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **make(name))
        print("wrote", name)
    np.savez_compressed(os.path.join(HERE, "ref_se3_table.npz"), **se3_table())
    print("wrote ref_se3_table")
