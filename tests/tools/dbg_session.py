import os, sys
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(_TESTS)); sys.path.insert(0, _TESTS)
import numpy as np, torch
from bundletrack_amd import synthetic as S
from bundletrack_amd.optimizer import OptimizerGpu, Workspace, BatchSolver, build_cache
from oracle import oracle as O
from helpers import OracleOptimizer, ParityOptimizer
from test_tracking_session import run_session
def main():
    dev = torch.device("cuda:0")
    ws = Workspace()
    par = ParityOptimizer(OptimizerGpu(workspace=ws), OracleOptimizer(O), S.pose_error)
    saved = {}
    orig = par.optimizeFrames
    n = [0]
    def wrap(*a):
        if n[0] in (22, 38, 7, 6): saved[n[0]] = (a[0].copy(), a[2], a[3], a[4], list(a[5]), list(a[7]), np.array(a[8], copy=True), a[9])
        n[0] += 1
        orig(*a)
    par.optimizeFrames = wrap
    seq, b, frames, errs = run_session(par, 45, to_device=lambda a: torch.from_numpy(a).to(dev))
    for call, (corr, K, H, W, depths, normals, poses, Kmat) in saved.items():
        campos, nrm, nvalid, intr = build_cache(ws, depths, normals, H, W, Kmat)
        ws.sync()
        cam_h, nrm_h = campos.cpu().numpy(), nrm.cpu().numpy()
        for mode in (1, 0):
            ref = O.solve(cam_h, nrm_h, intr, corr, poses, params=O.default_params(accum_mode=mode))
            bs = BatchSolver(ws)
            c, offs, mx = bs.pack_correspondences([corr], K)
            corr_d = torch.from_numpy(c.view(np.uint8).reshape(1, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
            poses_d = torch.from_numpy(poses[None].copy()).to(dev)
            tv = bs.trace_view(bs.solve(campos[None], nrm[None], intr, corr_d, offs_d, mx, poses_d, trace=True))
            print(f"call {call} accum_mode {mode}")
            for it in range(7):
                e = [S.pose_error(tv.T_after[0, it, k], ref.T_after[it, k]) for k in range(K)]
                cnt_g = tv.dense_pair[0, it, :, 27].astype(np.int64); cnt_o = ref.dense_count[it][:len(cnt_g)]
                print(f"  it{it} pose diff {max(max(x) for x in e):.2e} cnt diff {np.abs(cnt_g-cnt_o).max()} | alpha g {np.array2string(tv.pcg_scalars[0,it,:,1],precision=4)} o {np.array2string(ref.pcg_scalars[it,:,1],precision=4)} rz g {np.array2string(tv.pcg_scalars[0,it,:,0],precision=3)} o {np.array2string(ref.pcg_scalars[it,:,0],precision=3)}")
if __name__ == "__main__": main()
