"""A/B of fused / separate sweeps and tile counts at the bench workload (c3 x B, compact cache).  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bundletrack_amd import _lib
from bundletrack_amd.optimizer import BatchSolver, Workspace
import bench

def main():
    cfg = bench.CONFIGS["c3"]
    os.environ["BTBA_BENCH_NPROC"] = "1"
    inst = bench.generate_instances(cfg, [0, 1, 2, 3])
    dev = torch.device("cuda:0")
    ws = Workspace()
    for B in (int(a) for a in (sys.argv[1:] or ["32", "1"])):
        pick = [inst[b % 4] for b in range(B)]
        bs0 = BatchSolver(ws)
        corr, offs, mx = bs0.pack_correspondences([p["corr"] for p in pick], 15)
        zn_d = torch.from_numpy(np.stack([p["zn"] for p in pick])).to(dev)
        corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
        poses0 = torch.from_numpy(np.stack([p["poses"] for p in pick])).to(dev)
        ref = None
        for name, flag in (("no-fuse", 64), ("fuse", 128)):
            for tiles in ((3, 4, 5, 6, 8) if B > 1 else (5, 8, 10, 15)):
                bs = BatchSolver(ws, dense_tiles=tiles)
                bs.params.flags |= _lib.FLAG_TIME_KERNELS | flag
                poses_d = poses0.clone()
                for _ in range(2):
                    poses_d.copy_(poses0); bs.solve_zn(zn_d, pick[0]["H"], pick[0]["W"], pick[0]["K"], corr_d, offs_d, mx, poses_d)
                ws.sync(); ws.collect_stats()
                t0 = time.perf_counter()
                for _ in range(5):
                    poses_d.copy_(poses0); bs.solve_zn(zn_d, pick[0]["H"], pick[0]["W"], pick[0]["K"], corr_d, offs_d, mx, poses_d)
                ws.sync(); dt = (time.perf_counter() - t0) / 5
                st = ws.collect_stats()
                out = poses_d.cpu().numpy()
                if ref is None: ref = out
                print(f"B={B} {name} tiles={tiles}: step {dt*1e3:.3f} ms  dense {st['ms_dense_sweep']/st['n_dense_launches']*1e3:.1f} us/launch  sparse {st['ms_sparse_sweep']/max(st['n_sparse_launches'],1)*1e3:.1f}  sys {st['ms_system_solve']/st['n_solve_launches']*1e3:.1f}  maxdiff {np.abs(out-ref).max():.2e}", flush=True)


if __name__ == "__main__":
    main()
