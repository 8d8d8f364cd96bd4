"""A/B of the two per-instance system-solve kernels inside ONE process and ONE library: k_solve_small (BTBA_OPT_SOLVE_SMALL = 1, round 5)
against k_system_solve (0, rounds 1-4), alternating, on the bench workloads -- step time, average sweep / solve launch by hipEvents, and how far
the two kernels' final poses are apart.  Then the phase stamps of k_solve_small (trace clocks).  GPU box only.
    python scripts/ab_solve.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from bundletrack_amd import _lib, synthetic as S
from bundletrack_amd.optimizer import BatchSolver, Workspace


def setup(ws, cfg, inst, B, masked, dev):
    pick = [inst[b % len(inst)] for b in range(B)]
    K = cfg["K"]
    bs = BatchSolver(ws, weight_dense_depth=cfg["w_dense"])
    bs.params.flags |= _lib.FLAG_TIME_KERNELS | _lib.FLAG_TIME_SAMPLED | (_lib.FLAG_COMPACTION if masked else 0)
    corr, offs, mx = bs.pack_correspondences([p["corr"] for p in pick], K)
    zn_d = torch.from_numpy(np.stack([p["zn"] for p in pick])).to(dev)
    corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev)
    offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
    poses0 = torch.from_numpy(np.stack([p["poses"] for p in pick])).to(dev)
    poses_d = poses0.clone()
    aux = bs.cache_aux(zn_d, valid_lists=masked)
    use_c24 = not masked
    if use_c24:
        aux["corr24"] = bs.pack_correspondences24(corr_d, offs_d, mx, K)

    def step(trace=False):
        poses_d.copy_(poses0)
        return bs.solve_zn(zn_d, pick[0]["H"], pick[0]["W"], pick[0]["K"], None if use_c24 else corr_d, offs_d, mx, poses_d, aux=aux, corr_stride=corr_d.shape[1], trace=trace)
    return bs, step, poses_d


def measure(ws, step, poses_d, B, n):
    for _ in range(3):
        step()
    ws.sync(); ws.collect_stats()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    ws.sync()
    dt = (time.perf_counter() - t0) / n
    st = ws.collect_stats()
    return {"ms_per_step": round(dt * 1e3, 4), "git_per_s": round(B * 7 / dt, 0),
            "sweep_us": round((st["ms_dense_sweep"] / max(st["n_dense_launches"], 1) + st["ms_sparse_sweep"] / max(st["n_sparse_launches"], 1)) * 1e3, 2),
            "solve_us": round(st["ms_system_solve"] / max(st["n_solve_launches"], 1) * 1e3, 2), "tiles": st["dense_tiles"], "chunks": st["sparse_chunks"]}, poses_d.cpu().numpy().copy()


def main():
    os.environ.setdefault("BTBA_BENCH_NPROC", "8")
    dev = torch.device("cuda:0")
    ws = Workspace()
    out = {"what": "k_solve_small (opt 1) vs k_system_solve (opt 0), same library, alternating; hipEvents per launch (sampled iteration)", "rows": []}
    data = {}
    for name, cfgk, masked in (("c3", "c3", False), ("c3_masked", "c3", True), ("c2", "c2", False)):
        data[name] = (bench.CONFIGS[cfgk], bench.generate_instances(bench.CONFIGS[cfgk], list(range(8)), masked), masked)
    # settle the clocks
    cfg, inst, masked = data["c3"]
    bs, step, poses_d = setup(ws, cfg, inst, 32, masked, dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        step(); ws.sync()
    rows = () if os.environ.get("AB_PHASES_ONLY") else (("c3", 32, 40), ("c3_masked", 32, 60), ("c2", 32, 100), ("c3", 1, 100), ("c3_masked", 1, 100), ("c3", 8, 60), ("c3_masked", 8, 60))
    for name, B, n in rows:
        cfg, inst, masked = data[name]
        bs, step, poses_d = setup(ws, cfg, inst, B, masked, dev)
        res = {0: [], 1: []}
        poses = {}
        for opt in (0, 1, 0, 1, 0, 1):
            ws.set_option(_lib.OPT_SOLVE_SMALL, opt)
            r, p = measure(ws, step, poses_d, B, n)
            res[opt].append(r); poses[opt] = p
        worst_r = worst_t = 0.0
        for b in range(B):
            for k in range(cfg["K"]):
                rr, tt = S.pose_error(poses[0][b, k], poses[1][b, k])
                worst_r, worst_t = max(worst_r, rr), max(worst_t, tt)
        row = {"workload": name, "B": B, "legacy": res[0], "small": res[1], "pose_diff_rot": float(f"{worst_r:.3e}"), "pose_diff_trans": float(f"{worst_t:.3e}"),
               "finite": bool(np.isfinite(poses[1]).all())}
        best = lambda rs, k: min(x[k] for x in rs)
        row["summary"] = {"ms_per_step": [best(res[0], "ms_per_step"), best(res[1], "ms_per_step")], "solve_us": [best(res[0], "solve_us"), best(res[1], "solve_us")]}
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
    # phase stamps of k_solve_small (shader-clock cycles from the kernel's start; with the trace dumps in between 3 and 4)
    ws.set_option(_lib.OPT_SOLVE_SMALL, 1)
    out["phases"] = []
    for name, B in (("c3", 1), ("c3", 32), ("c3_masked", 1)):
        cfg, inst, masked = data[name]
        bs, step, poses_d = setup(ws, cfg, inst, B, masked, dev)
        bs.params.flags &= ~(_lib.FLAG_TIME_KERNELS | _lib.FLAG_TIME_SAMPLED)
        tv = bs.trace_view(step(trace=True))
        c = tv.clk.mean(axis=(0, 1))                     # [B, n_gn, 8]
        ph = {"workload": name, "B": B, "loads_issued": float(c[0]), "staged": float(c[1] - c[0]), "offdiag_framesums": float(c[2] - c[1]), "expand": float(c[3] - c[2]),
              "trace_dump": float(c[4] - c[3]), "pcg": float(c[5] - c[4]), "update": float(c[6] - c[5]), "inverse": float(c[7] - c[6]), "total_wo_dump": float(c[7] - (c[4] - c[3]))}
        out["phases"].append(ph)
        print(json.dumps(ph), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
