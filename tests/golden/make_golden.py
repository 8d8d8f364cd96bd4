"""Generates tests/golden/*.npz.

SELF-DERIVED regression fixtures: outputs of THIS repo's CPU oracle (oracle/btba_oracle.c, accum_mode=1) on seeded
synthetic problems; they pin the oracle against regressions and give the GPU tests a fixed target.  The golden vectors
produced by the REFERENCE itself are the ref_*.npz files next to these (make_reference_golden.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bundletrack_amd import synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    # name: (n_frames, corr/pair, seed, background, weight_dense, weight_sparse)
    "k3_sparse_dense_bg": (3, 80, 101, True, 1.0, 1.0),
    "k4_sparse_dense_masked": (4, 60, 102, False, 1.0, 1.0),
    "k4_sparse_only": (4, 100, 103, True, 0.0, 1.0),
}


def make(name):
    n, m, seed, bg, wd, ws = CASES[name]
    pb = S.make_problem(n, m, seed, background=bg, H=120, W=160, downscale=4,
                        K=S.NOCS_K * np.array([[0.25], [0.25], [1.0]]))
    caches = [O.build_cache(pb.depth[k], pb.normals[k], pb.K, 4.0) for k in range(n)]
    campos = np.stack([c["campos"] for c in caches])
    normals = np.stack([c["normals"] for c in caches])
    prm = O.default_params(weight_dense_depth=wd, weight_sparse=ws, n_threads=1)
    tr = O.solve(campos, normals, caches[0]["intr"], pb.corr, pb.poses_init, params=prm)
    return dict(depth=pb.depth, normals_full=pb.normals, K=pb.K, campos=campos, normals=normals, intr=caches[0]["intr"],
                corr=pb.corr.view(np.uint8).reshape(-1, 32), n_match_per_pair=pb.n_match_per_pair, poses_init=pb.poses_init,
                weight_dense=np.float32(wd), weight_sparse=np.float32(ws),
                T_after=tr.T_after, x_after=tr.x_after, dense_count=tr.dense_count, pcg_scalars=tr.pcg_scalars,
                rhs=tr.rhs, precond=tr.precond, poses_out=tr.poses)


def make_xorwow():
    """tables/xorwow_seed0_first64.npz -- SELF-DERIVED (oracle/xorwow.h): the first three curand() / curand_uniform() draws of
    curand_init(0, t, 0) for t = 0 .. 63, as this repository computes them.  It pins the product and the oracle against change,
    and it is the table to hold a CUDA machine's output against ("checking the cuRAND constants")."""
    from oracle import oracle
    u = oracle.ransac_reference_uniforms(64)
    raw = np.stack([oracle.curand_xorwow_draw(0, t, 0, 3)[0] for t in range(64)])
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tables", "xorwow_seed0_first64.npz"), uniforms=u, raw=raw)


if __name__ == "__main__":
    for name in CASES:
        d = make(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, {k: v.shape for k, v in d.items() if hasattr(v, "shape") and v.ndim}, os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KiB")
    make_xorwow()
