"""Phase timing of k_system_solve from the trace's clock stamps (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bundletrack_amd.optimizer import BatchSolver, Workspace
import bench


def main():
    os.environ["BTBA_BENCH_NPROC"] = "1"
    cfg = bench.CONFIGS["c3"]
    inst = bench.generate_instances(cfg, [0, 1])
    dev = torch.device("cuda:0")
    ws = Workspace()
    for B in (1, 32):
        pick = [inst[b % 2] for b in range(B)]
        bs = BatchSolver(ws)
        corr, offs, mx = bs.pack_correspondences([p["corr"] for p in pick], 15)
        cam_d = torch.from_numpy(np.stack([p["campos"] for p in pick])).to(dev); nrm_d = torch.from_numpy(np.stack([p["normals"] for p in pick])).to(dev)
        corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
        poses_d = torch.from_numpy(np.stack([p["poses"] for p in pick])).to(dev)
        tv = bs.trace_view(bs.solve(cam_d, nrm_d, pick[0]["intr"], corr_d, offs_d, mx, poses_d, trace=True))
        clk = tv.clk[0]          # [n_gn, 8]
        d = np.diff(np.concatenate([np.zeros((clk.shape[0], 1)), clk[:, :5]], 1), axis=1)
        print(f"B={B}: k_system_solve phase shader-clock cycles (mean over GN iterations): reduce {d[:,0].mean():.0f}  congruence {d[:,1].mean():.0f}  assemble(+trace dump) {d[:,2].mean():.0f}  PCG {d[:,3].mean():.0f}  update {d[:,4].mean():.0f}  total {clk[:,4].mean():.0f}")


if __name__ == "__main__":
    main()
