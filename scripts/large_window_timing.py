"""Solve time of windows around and above BTBA_MAX_FRAMES_LDS (31 frames: matrix in LDS, one-wave PCG) up to BTBA_MAX_FRAMES
(85: matrix in a global scratch, 16-wave PCG), through the drop-in boundary on 640x480 frames.  GPU box only.
    python scripts/large_window_timing.py > gpurun_out/large_window_timing.jsonl"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bundletrack_amd import _lib, synthetic as S
from bundletrack_amd.optimizer import OptimizerGpu, Workspace


def main():
    dev = torch.device("cuda:0")
    ws = Workspace()
    for (N, m) in ((15, 500), (31, 500), (32, 500), (40, 500), (60, 300), (85, 200)):
        pb = S.make_problem(N, m, seed=700 + N, background=False, rot_step_deg=(4.0, 5.0))
        depths = [torch.from_numpy(pb.depth[k]).to(dev) for k in range(N)]
        normals = [torch.from_numpy(pb.normals[k]).to(dev) for k in range(N)]
        opt = OptimizerGpu(workspace=ws)
        walls = []
        keys = list(range(N))
        for timed in (False, True):
            if timed: opt.params.flags |= _lib.FLAG_TIME_KERNELS
            for rep in range(12):
                poses = pb.poses_init.copy()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                opt.optimizeFrames(pb.corr, pb.n_match_per_pair, N, pb.H, pb.W, depths, None, normals, poses, pb.K, frame_keys=keys)
                dt = time.perf_counter() - t0
                if not timed and rep >= 4: walls.append(dt)
        st = opt.last_stats
        walls = np.array(walls) * 1e3
        err = max(max(S.pose_error(poses[k], pb.poses_gt[k])) for k in range(N))
        print(json.dumps(dict(N=N, corr_per_pair=m, n_corr=int(len(pb.corr)), pairs=N * (N - 1) // 2, matrix="lds" if N <= 31 else "global",
                              wall_ms_median_cached_frames=round(float(np.median(walls)), 3), gn_iters_per_s=round(7e3 / float(np.median(walls)), 1),
                              stats_ms={k: round(float(v), 4) for k, v in st.items() if k.startswith("ms_")}, err_vs_gt=float(err))), flush=True)


if __name__ == "__main__":
    main()
