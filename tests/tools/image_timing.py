"""Per-frame pre-processing timing at 640x480: btba_process_depth (erode + 2 bilateral passes,
one launch) and btba_depth_to_normals (one launch), against the CPU oracle.  GPU box only."""
import json, os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(_TESTS)); sys.path.insert(0, _TESTS)
import numpy as np, torch
from bundletrack_amd import synthetic as S
from bundletrack_amd.optimizer import Workspace, process_depth, depth_to_normals, DEPTH_PROCESSING_DEFAULTS
from oracle import oracle as O


def gpu_time(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    ws = Workspace()
    pb = S.make_problem(2, 10, seed=5, background=True)
    rng = np.random.default_rng(0)
    depth = (pb.depth[0] + rng.normal(scale=0.001, size=pb.depth[0].shape)).astype(np.float32)
    d = torch.from_numpy(depth).to(dev)
    H, W = depth.shape
    p = DEPTH_PROCESSING_DEFAULTS
    t_pd = gpu_time(lambda: process_depth(ws, d))
    filt = process_depth(ws, d)
    t_nm = gpu_time(lambda: depth_to_normals(ws, filt, pb.K))
    t0 = time.perf_counter(); O.process_depth(depth, **p); c_pd = time.perf_counter() - t0
    t0 = time.perf_counter(); O.depth_to_normals(filt.cpu().numpy(), pb.K); c_nm = time.perf_counter() - t0
    taps = (2 * p["erode_radius"] + 1) ** 2 + 2 * 2 * (2 * p["bf_radius"] + 1) ** 2
    print(json.dumps(dict(frame=f"{W}x{H}", params=p,
                          process_depth=dict(gpu_us=round(t_pd * 1e3, 1), algorithmic_bytes=8 * H * W, GBps=round(8 * H * W / (t_pd * 1e-3) / 1e9, 1),
                                             window_taps_per_pixel=taps, gtaps_per_s=round(taps * H * W / (t_pd * 1e-3) / 1e9, 1), cpu_oracle_ms=round(c_pd * 1e3, 1)),
                          depth_to_normals=dict(gpu_us=round(t_nm * 1e3, 1), algorithmic_bytes=20 * H * W, GBps=round(20 * H * W / (t_nm * 1e-3) / 1e9, 1),
                                                cpu_oracle_ms=round(c_nm * 1e3, 1)))))


if __name__ == "__main__":
    main()
