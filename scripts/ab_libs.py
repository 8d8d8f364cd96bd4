"""A/B of kernel builds at the bench workload.  GPU box only.
    python scripts/ab_libs.py build/ab/A.so build/ab/B.so ...        (parent: generates the instances once, one child per library)
Each child loads ONE library (BTBA_LIB_PATH), runs c3 x 32 on 100 %-valid and on masked frames, prints step time, the fused
sweep's and the system solve's average launch time and a checksum of the poses."""
import json, os, pickle, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CACHE = "/tmp/ab_instances_%s.pkl" % os.environ.get("AB_CONFIG", "c3")          # AB_CONFIG=c4: K = 30, 4 000 correspondences per pair (full frames only)


def child(lib):
    import numpy as np, torch
    from bundletrack_amd import _lib
    from bundletrack_amd.optimizer import BatchSolver, Workspace
    data = pickle.load(open(CACHE, "rb"))
    dev = torch.device("cuda:0")
    ws = Workspace()
    out = {"lib": os.path.basename(lib)}
    for tag, inst in data.items():
        B = int(os.environ.get("AB_B", "32"))
        pick = [inst[(b % len(inst)) % int(os.environ.get("AB_DISTINCT", "8"))] for b in range(B)]        # AB_DISTINCT=1: a batch of identical instances (l2 alias probe)
        bs = BatchSolver(ws)
        bs.params.flags |= (0 if os.environ.get("AB_NO_TIMING") else _lib.FLAG_TIME_KERNELS) | (_lib.FLAG_COMPACTION if tag == "masked" else 0) | int(os.environ.get("AB_FLAGS", "0"))
        bs.params.dense_tiles = int(os.environ.get("BTBA_BENCH_TILES", "0"))
        bs.params.reduction_mode = int(os.environ.get("AB_REDUCTION", "0"))
        corr, offs, mx = bs.pack_correspondences([p["corr"] for p in pick], int(pick[0]["poses"].shape[0]))
        zn_d = torch.from_numpy(np.stack([p["zn"] for p in pick])).to(dev)
        corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
        poses0 = torch.from_numpy(np.stack([p["poses"] for p in pick])).to(dev)
        poses_d = poses0.clone()
        def step():
            poses_d.copy_(poses0); bs.solve_zn(zn_d, pick[0]["H"], pick[0]["W"], pick[0]["K"], corr_d, offs_d, mx, poses_d)
        for _ in range(3): step()
        ws.sync(); ws.collect_stats()
        n = 12
        t0 = time.perf_counter()
        for _ in range(n): step()
        ws.sync(); dt = (time.perf_counter() - t0) / n
        st = ws.collect_stats()
        p = poses_d.cpu().numpy()
        out[tag] = {"ms_per_step": round(dt * 1e3, 4), "sweep_us": round(st["ms_dense_sweep"] / max(st["n_dense_launches"], 1) * 1e3, 2),
                    "solve_us": round(st["ms_system_solve"] / max(st["n_solve_launches"], 1) * 1e3, 2), "tiles": st["dense_tiles"],
                    "git_per_s": round(B * 7 / dt, 0), "checksum": float(np.abs(p).sum()), "finite": bool(np.isfinite(p).all())}
    print(json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    import bench
    os.environ.setdefault("BTBA_BENCH_NPROC", "8")
    cfg = bench.CONFIGS[os.environ.get("AB_CONFIG", "c3")]
    if not os.path.exists(CACHE):
        data = {"full": bench.generate_instances(cfg, list(range(8)))}
        if os.environ.get("AB_CONFIG", "c3") == "c3":
            data["masked"] = bench.generate_instances(cfg, list(range(8)), masked=True)
        pickle.dump(data, open(CACHE, "wb"))
    for spec in sys.argv[1:]:
        lib, *sets = spec.split(":")                       # build/ab/x.so:BTBA_NO_PERSISTENT=1 -- environment of that child only
        env = dict(os.environ, BTBA_LIB_PATH=os.path.abspath(lib))
        env.update(dict(kv.split("=", 1) for kv in sets))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], env=env, capture_output=True, text=True, timeout=300)
        print((r.stdout.strip()[:-1] + ', "env": ' + json.dumps(sets) + "}") if r.stdout.strip() else ("FAILED " + lib + " " + r.stderr[-600:]), flush=True)


if __name__ == "__main__":
    main()
