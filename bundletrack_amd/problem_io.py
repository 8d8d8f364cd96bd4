"""Problem dumps: one bundle-adjustment call -- what Bundler::optimizeGPU hands to OptimizerGpu::optimizeFrames
(src/Bundler.cpp:286-351) -- as one little-endian binary file, so that the HIP path, the CPU oracle, the C++ host layer
(btba::loadProblem, bundletrack_amd/cpp/btba_host.hpp) and a reference build consume identical bytes.

    offset  type                     field
    0       char[8]                  magic "BTBAPRB1"
    8       int32                    n_frames N
    12      int32                    H                      full-resolution rows
    16      int32                    W                      full-resolution columns
    20      uint32                   n_corr C
    24      int32                    flags                  bit 0: poses_gt present
    28      float32                  image_downscale        bundle.image_downscale (config_ycbineoat.yml:31)
    32      float32[9]               K                      row-major full-resolution intrinsics
    68      EntryJ[C]                corr                   32 B each (SIFTImageManager.h:44-59), pair-major as optimizeGPU emits them
    ...     int32[N(N-1)/2]          n_match_per_pair       segment lengths in pair order (Bundler.cpp:322)
    ...     float32[N][16]           poses_init             camera->model, row-major (what LossGPU.cu:88-97 uploads)
    ...     float64[N][16]           poses_gt               only with flags bit 0 (synthetic problems)
    ...     float32[N][H][W]         depth                  metres, 0 = invalid (Frame.h:73)
    ...     float32[N][H][W][4]      normals                xyz unit, w = 0, zeros = invalid (Frame.h:75)
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ._lib import ENTRYJ_DTYPE

MAGIC = b"BTBAPRB1"
FLAG_HAS_GT = 1


@dataclass
class ProblemDump:
    K: np.ndarray                  # [3,3] f32
    H: int
    W: int
    image_downscale: float
    corr: np.ndarray               # ENTRYJ_DTYPE[C]
    n_match_per_pair: np.ndarray   # [P] i32
    poses_init: np.ndarray         # [N,4,4] f32
    poses_gt: np.ndarray | None    # [N,4,4] f64
    depth: np.ndarray              # [N,H,W] f32
    normals: np.ndarray            # [N,H,W,4] f32

    @property
    def n_frames(self) -> int:
        return int(self.poses_init.shape[0])


def save_problem(path: str, pb) -> None:
    """`pb`: a ProblemDump or a synthetic.Problem rendered at full resolution (depth / normals present)."""
    if pb.depth is None or pb.normals is None:
        raise ValueError("a problem dump carries the full-resolution frames: make_problem(..., full_res=True)")
    N = int(pb.poses_init.shape[0])
    H, W = int(pb.H), int(pb.W)
    corr = np.ascontiguousarray(pb.corr, ENTRYJ_DTYPE)
    nm = np.ascontiguousarray(pb.n_match_per_pair, "<i4")
    if nm.shape != (N * (N - 1) // 2,):
        raise ValueError("n_match_per_pair needs one length per frame pair")
    depth = np.ascontiguousarray(pb.depth, "<f4")
    normals = np.ascontiguousarray(pb.normals, "<f4")
    if depth.shape != (N, H, W) or normals.shape != (N, H, W, 4):
        raise ValueError("depth must be [N,H,W] and normals [N,H,W,4]")
    gt = getattr(pb, "poses_gt", None)
    scale = float(getattr(pb, "image_downscale", getattr(pb, "downscale", 4)))
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(np.array([N, H, W], "<i4").tobytes())
        f.write(np.array([corr.shape[0]], "<u4").tobytes())
        f.write(np.array([FLAG_HAS_GT if gt is not None else 0], "<i4").tobytes())
        f.write(np.array([scale], "<f4").tobytes())
        f.write(np.ascontiguousarray(pb.K, "<f4").reshape(9).tobytes())
        f.write(corr.tobytes())
        f.write(nm.tobytes())
        f.write(np.ascontiguousarray(pb.poses_init, "<f4").reshape(N, 16).tobytes())
        if gt is not None:
            f.write(np.ascontiguousarray(gt, "<f8").reshape(N, 16).tobytes())
        f.write(depth.tobytes())
        f.write(normals.tobytes())


def load_problem(path: str) -> ProblemDump:
    buf = np.fromfile(path, np.uint8)
    if buf.size < 68 or buf[:8].tobytes() != MAGIC:
        raise ValueError(f"{path}: not a BTBAPRB1 problem dump")
    N, H, W = (int(v) for v in buf[8:20].view("<i4"))
    C = int(buf[20:24].view("<u4")[0])
    flags = int(buf[24:28].view("<i4")[0])
    scale = float(buf[28:32].view("<f4")[0])
    if N < 1 or H < 1 or W < 1:
        raise ValueError(f"{path}: corrupt header")
    P = N * (N - 1) // 2
    sizes = [36, 32 * C, 4 * P, 64 * N, 128 * N if flags & FLAG_HAS_GT else 0, 4 * N * H * W, 16 * N * H * W]
    if buf.size != 32 + sum(sizes):
        raise ValueError(f"{path}: {buf.size} bytes, the header announces {32 + sum(sizes)}")
    off = np.concatenate([[32], 32 + np.cumsum(sizes)]).astype(np.int64)
    part = [buf[off[k]:off[k + 1]] for k in range(len(sizes))]
    return ProblemDump(
        K=part[0].view("<f4").reshape(3, 3).copy(), H=H, W=W, image_downscale=scale,
        corr=part[1].view(ENTRYJ_DTYPE).copy(), n_match_per_pair=part[2].view("<i4").copy(),
        poses_init=part[3].view("<f4").reshape(N, 4, 4).copy(),
        poses_gt=part[4].view("<f8").reshape(N, 4, 4).copy() if flags & FLAG_HAS_GT else None,
        depth=part[5].view("<f4").reshape(N, H, W).copy(), normals=part[6].view("<f4").reshape(N, H, W, 4).copy())


def _main(argv):
    """python -m bundletrack_amd.problem_io info <dump>            header and sizes
       python -m bundletrack_amd.problem_io solve <dump> [out.txt]   run btba_optimize_frames on it (needs the GPU); poses as 4x4 text"""
    import sys
    if len(argv) >= 2 and argv[0] == "info":
        pb = load_problem(argv[1])
        print(f"{argv[1]}: {pb.n_frames} frames {pb.W}x{pb.H}, downscale {pb.image_downscale:g}, {len(pb.corr)} correspondences "
              f"(pair segments {int(pb.n_match_per_pair.min()) if pb.n_match_per_pair.size else 0}..{int(pb.n_match_per_pair.max()) if pb.n_match_per_pair.size else 0}), "
              f"{float((pb.depth >= 0.1).mean()) * 100:.1f} % valid depth, ground truth {'yes' if pb.poses_gt is not None else 'no'}")
        return 0
    if len(argv) >= 2 and argv[0] == "solve":
        import torch
        from .optimizer import OptimizerGpu, Workspace
        pb = load_problem(argv[1])
        dev = torch.device("cuda:0")
        depths = [torch.from_numpy(pb.depth[k]).to(dev) for k in range(pb.n_frames)]
        normals = [torch.from_numpy(pb.normals[k]).to(dev) for k in range(pb.n_frames)]
        opt = OptimizerGpu(workspace=Workspace(), image_downscale=pb.image_downscale)
        poses = pb.poses_init.copy()
        opt.optimizeFrames(pb.corr, pb.n_match_per_pair, pb.n_frames, pb.H, pb.W, depths, None, normals, poses, pb.K)
        text = "\n\n".join("\n".join(" ".join(f"{v:.10g}" for v in row) for row in P) for P in poses)
        if len(argv) >= 3:
            open(argv[2], "w").write(text + "\n")
        else:
            print(text)
        print(f"solve {opt.last_stats['ms_solve']:.3f} ms, call {opt.last_stats['ms_total']:.3f} ms", file=sys.stderr)
        return 0
    print(_main.__doc__, file=sys.stderr)
    return 1


if __name__ == "__main__":
    import sys
    sys.exit(_main(sys.argv[1:]))
