"""GPU diagnostic: HIP path vs CPU oracle, quantity by quantity, plus a first timing.
Run on the GPU box:  python tests/tools/gpu_diag.py [K] [m]"""
import os
import sys
import time

_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(_TESTS)); sys.path.insert(0, _TESTS)
import numpy as np
import torch

from bundletrack_amd import _lib, synthetic as S
from bundletrack_amd.optimizer import BatchSolver, OptimizerGpu, Workspace, build_cache
from oracle import oracle as O

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
print("device", torch.cuda.get_device_name(0))
pb = S.make_problem(K, m, seed=11, background=True)
depths = [torch.from_numpy(pb.depth[k]).to(dev) for k in range(K)]
normals = [torch.from_numpy(pb.normals[k]).to(dev) for k in range(K)]
ws = Workspace()

# ---- cache build: bit-exact vs oracle
campos, nrm, nvalid, intr = build_cache(ws, depths, normals, pb.H, pb.W, pb.K)
ws.sync()
oc = [O.build_cache(pb.depth[k], pb.normals[k], pb.K) for k in range(K)]
ocam = np.stack([c["campos"] for c in oc]); onrm = np.stack([c["normals"] for c in oc])
print("cache campos bit-exact:", np.array_equal(campos.cpu().numpy().view(np.uint32), ocam.view(np.uint32)),
      "max abs diff", np.abs(campos.cpu().numpy() - ocam).max(),
      "normals bit-exact:", np.array_equal(nrm.cpu().numpy().view(np.uint32), onrm.view(np.uint32)),
      "n_valid", nvalid.cpu().numpy().tolist(), [c["n_valid"] for c in oc], "intr", intr, oc[0]["intr"])

# ---- batch solve with trace
ref = O.solve(ocam, onrm, oc[0]["intr"], pb.corr, pb.poses_init)
bs = BatchSolver(ws)
corr, offs, mx = bs.pack_correspondences([pb.corr], K)
corr_d = torch.from_numpy(corr.view(np.uint8).reshape(1, -1, 32)).to(dev)
offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
poses_d = torch.from_numpy(pb.poses_init[None].copy()).to(dev)
tr = bs.solve(campos[None], nrm[None], intr, corr_d, offs_d, mx, poses_d, trace=True)
tv = bs.trace_view(tr)
P = K * (K - 1) // 2
pairs = O.target_lower_pairs(K)
for it in range(bs.params.n_gn_iters):
    cnt_g = tv.dense_pair[0, it, :, 27].astype(np.int64)
    cnt_o = ref.dense_count[it][:P]
    # dense JtJ block check: S for pair p sits at rows/cols of frame j (diag) -- compare the cross block (j rows, i cols) = -S
    dS = 0.0
    for p, (i, j) in enumerate(pairs):
        if i == 0:
            continue
        S21 = tv.dense_pair[0, it, p, :21]
        Sm = np.zeros((6, 6), np.float32)
        k = 0
        for r in range(6):
            for c in range(r, 6):
                Sm[r, c] = Sm[c, r] = S21[k]; k += 1
        blk = ref.dense_JtJ[it][6 * j:6 * j + 6, 6 * i:6 * i + 6]
        dS = max(dS, np.abs(blk + Sm).max() / max(1e-12, np.abs(blk).max()))
    rhs_err = np.abs(tv.rhs[0, it] - ref.rhs[it]).max() / max(1e-12, np.abs(ref.rhs[it]).max())
    pre_err = np.abs(tv.precond[0, it] - ref.precond[it]).max() / max(1e-12, np.abs(ref.precond[it]).max())
    errs = [S.pose_error(tv.T_after[0, it, k], ref.T_after[it, k]) for k in range(K)]
    print(f"it{it}: dense cnt diff max {np.abs(cnt_g - cnt_o).max()} (tot {cnt_o.sum()}) S relerr {dS:.2e} rhs relerr {rhs_err:.2e} prec relerr {pre_err:.2e}"
          f" | pose diff rot {max(e[0] for e in errs):.2e} trans {max(e[1] for e in errs):.2e}")
    print("    alpha gpu", tv.pcg_scalars[0, it, :, 1], "\n    alpha ora", ref.pcg_scalars[it, :, 1])
ws.sync()
fin = poses_d.cpu().numpy()[0]
errs = [S.pose_error(fin[k], ref.poses[k]) for k in range(K)]
print("final pose diff rot %.3e trans %.3e" % (max(e[0] for e in errs), max(e[1] for e in errs)))

# ---- drop-in boundary
poses = pb.poses_init.copy()
opt = OptimizerGpu(workspace=ws)
opt.params.flags |= _lib.FLAG_TIME_KERNELS
opt.optimizeFrames(pb.corr, pb.n_match_per_pair, K, pb.H, pb.W, depths, None, normals, poses, pb.K)
errs = [S.pose_error(poses[k], ref.poses[k]) for k in range(K)]
print("optimizeFrames pose diff rot %.3e trans %.3e" % (max(e[0] for e in errs), max(e[1] for e in errs)), opt.last_stats)
for _ in range(3):
    poses = pb.poses_init.copy()
    opt.optimizeFrames(pb.corr, pb.n_match_per_pair, K, pb.H, pb.W, depths, None, normals, poses, pb.K)
    print("  again:", {k: v for k, v in opt.last_stats.items() if k.startswith("ms_")})

# ---- timing at c3 size (K=15, 2000/pair), single + batch
if os.environ.get("BTBA_DIAG_TIMING", "1") == "1":
    for (Kc, mc, B) in ((15, 2000, 1), (15, 2000, 8), (15, 2000, 32)):
        pbs = [S.make_problem(Kc, mc, seed=S.config_seed(3, b), background=True, full_res=False) for b in range(min(B, 4))]
        from bundletrack_amd.synthetic import cache_source_pixels
        intr_c = None
        cams, nrms = [], []
        for pbk in pbs:
            # cache from the directly rendered cache-resolution depth: campos via oracle on a fake "full-res == cache-res" is not
            # needed here -- build campos analytically with the oracle's cache builder on the upsampled grid is costly; use GPU path:
            Hd, Wd = pbk.cache_depth.shape[1:]
            xi, yi = cache_source_pixels(pbk.H, pbk.W, Hd, Wd)
            Kf = pbk.K.astype(np.float64)
            d = pbk.cache_depth.astype(np.float32)
            x = ((xi[None, None, :].astype(np.float32) * d) - np.float32(Kf[0, 2]) * d) / np.float32(Kf[0, 0])
            y = ((yi[None, :, None].astype(np.float32) * d) - np.float32(Kf[1, 2]) * d) / np.float32(Kf[1, 1])
            cp = np.stack([x, y, d, np.ones_like(d)], -1).astype(np.float32)
            cp[d < 0.1] = 0
            cams.append(cp); nrms.append(pbk.cache_normals)
        reps = (B + len(pbs) - 1) // len(pbs)
        cam_d = torch.from_numpy(np.stack((cams * reps)[:B])).to(dev)
        nrm_d = torch.from_numpy(np.stack((nrms * reps)[:B])).to(dev)
        corr, offs, mx = bs.pack_correspondences(([p.corr for p in pbs] * reps)[:B], Kc)
        corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev)
        offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
        p0 = torch.from_numpy(np.stack(([p.poses_init for p in pbs] * reps)[:B])).to(dev)
        intr_c = np.array([pbs[0].K[0, 0] * (Wd / pbs[0].W), pbs[0].K[1, 1] * (Hd / pbs[0].H), pbs[0].K[0, 2] * ((Wd - 1) / (pbs[0].W - 1)), pbs[0].K[1, 2] * ((Hd - 1) / (pbs[0].H - 1))], np.float32)
        bs.params.flags |= _lib.FLAG_TIME_KERNELS
        for rep in range(3):
            poses_d = p0.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bs.solve(cam_d, nrm_d, intr_c, corr_d, offs_d, mx, poses_d)
            ws.sync()
            dt = time.perf_counter() - t0
            st = ws.collect_stats()
        gn = B * bs.params.n_gn_iters
        print(f"K={Kc} m={mc} B={B}: wall {dt*1e3:.3f} ms -> {gn/dt:.0f} GN it/s | ms_solve {st['ms_solve']:.3f} dense {st['ms_dense_sweep']:.3f}/{st['n_dense_launches']}"
              f" sparse {st['ms_sparse_sweep']:.3f}/{st['n_sparse_launches']} sys {st['ms_system_solve']:.3f}/{st['n_solve_launches']} tiles {st['dense_tiles']} chunks {st['sparse_chunks']}"
              f" | dense alg GB/s {st['bytes_dense_alg']*st['n_dense_launches']/max(st['ms_dense_sweep'],1e-9)/1e6:.0f}")
        fin = poses_d.cpu().numpy()
        e = [S.pose_error(fin[0, k], pbs[0].poses_gt[k]) for k in range(Kc)]
        print("   err vs GT rot %.2e trans %.2e finite %s" % (max(x[0] for x in e), max(x[1] for x in e), np.isfinite(fin).all()))
print("DIAG DONE")
