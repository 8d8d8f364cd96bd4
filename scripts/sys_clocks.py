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
        c = clk.mean(0)
        print(f"B={B}: k_system_solve phase shader-clock cycles (mean over GN iterations): [wave 0: staging {c[6]:.0f}, partial sums {c[7]-c[6]:.0f}, zero-fill + barrier {c[0]-c[7]:.0f}] reduce {c[0]:.0f}  congruence {c[1]-c[0]:.0f}  assemble {c[2]-c[1]:.0f}  [trace dump {c[5]-c[2]:.0f}]  PCG {c[3]-c[5]:.0f}  update {c[4]-c[3]:.0f}  total w/o dump {c[4]-(c[5]-c[2]):.0f}")


if __name__ == "__main__":
    main()
