"""Phase stamps of k_system_solve on large windows (matrix in global scratch).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bundletrack_amd import synthetic as S
from bundletrack_amd.optimizer import BatchSolver, Workspace


def main():
    dev = torch.device("cuda:0")
    ws = Workspace()
    for (N, m) in ((31, 500), (40, 500), (85, 200)):
        pb = S.make_problem(N, m, seed=700 + N, background=False, rot_step_deg=(4.0, 5.0), full_res=False)
        bs = BatchSolver(ws)
        corr, offs, mx = bs.pack_correspondences([pb.corr], N)
        zn_d = torch.from_numpy(S.compact_cache(pb)[None]).to(dev)
        corr_d = torch.from_numpy(corr.view(np.uint8).reshape(1, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
        poses_d = torch.from_numpy(pb.poses_init[None].copy()).to(dev)
        tv = bs.trace_view(bs.solve_zn(zn_d, pb.H, pb.W, pb.K, corr_d, offs_d, mx, poses_d, trace=True))
        c = tv.clk[0].mean(0)
        print(f"N={N}: staging {c[6]:.0f}, partial sums {c[7]-c[6]:.0f}, stores+zero-fill+barrier {c[0]-c[7]:.0f} | reduce {c[0]:.0f}  assemble {c[2]-c[1]:.0f}  [trace dump {c[5]-c[2]:.0f}]  PCG {c[3]-c[5]:.0f}  update {c[4]-c[3]:.0f}  total w/o dump {c[4]-(c[5]-c[2]):.0f} cycles", flush=True)


if __name__ == "__main__":
    main()
