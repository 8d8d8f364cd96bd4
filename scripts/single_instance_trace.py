"""One c3 instance (K=15, 2k corr/pair, 100 %-valid frames unless --masked) solved N times back to back, for a
`rocprofv3 --kernel-trace` timeline of the single-window case (the tracker's own use: B = 1).  GPU box only.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_b1 -o b1 -- python scripts/single_instance_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bundletrack_amd.optimizer import BatchSolver, Workspace
import bench


def main():
    masked = "--masked" in sys.argv
    os.environ["BTBA_BENCH_NPROC"] = "1"
    inst = bench.generate_instances(bench.CONFIGS["c3"], [0], masked=masked)
    dev = torch.device("cuda:0")
    ws = Workspace()
    bs = BatchSolver(ws)
    p = inst[0]
    corr, offs, mx = bs.pack_correspondences([p["corr"]], 15)
    zn_d = torch.from_numpy(p["zn"][None]).to(dev)
    corr_d = torch.from_numpy(corr.view(np.uint8).reshape(1, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
    poses0 = torch.from_numpy(p["poses"][None]).to(dev)
    poses_d = poses0.clone()
    for rep in range(3):
        n = 30 if rep == 0 else 100
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            bs.solve_zn(zn_d, p["H"], p["W"], p["K"], corr_d, offs_d, mx, poses_d)
        ws.sync(); dt = (time.perf_counter() - t0) / n
        print(f"single instance{' (masked)' if masked else ''}: {dt * 1e3:.4f} ms per solve, {7 / dt:.0f} GN it/s", flush=True)


if __name__ == "__main__":
    main()
