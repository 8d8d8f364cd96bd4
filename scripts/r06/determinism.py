"""Developer probe: is a build deterministic, and does it give the bits of another?   python scripts/r06/determinism.py lib1.so lib2.so ...
(each library in its own child process: c3 x 8 on 100 %-valid and on masked frames, three solves each, per-instance checksums)"""
import json, os, pickle, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CACHE = "/tmp/ab_instances.pkl"


def child(lib):
    import numpy as np, torch
    from bundletrack_amd import _lib
    from bundletrack_amd.optimizer import BatchSolver, Workspace
    data = pickle.load(open(CACHE, "rb"))
    dev = torch.device("cuda:0")
    ws = Workspace()
    out = {"lib": os.path.basename(lib)}
    B = int(os.environ.get("AB_B", "8"))
    for tag, inst in data.items():
        pick = [inst[b % len(inst)] for b in range(B)]
        bs = BatchSolver(ws)
        bs.params.flags |= (_lib.FLAG_COMPACTION if tag == "masked" else 0)
        bs.params.dense_tiles = int(os.environ.get("BTBA_BENCH_TILES", "0"))
        corr, offs, mx = bs.pack_correspondences([p["corr"] for p in pick], 15)
        zn_d = torch.from_numpy(np.stack([p["zn"] for p in pick])).to(dev)
        corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
        poses0 = torch.from_numpy(np.stack([p["poses"] for p in pick])).to(dev)
        sums = []
        for rep in range(3):
            poses_d = poses0.clone()
            bs.solve_zn(zn_d, pick[0]["H"], pick[0]["W"], pick[0]["K"], corr_d, offs_d, mx, poses_d)
            ws.sync()
            p = poses_d.cpu().numpy().astype(np.float64)
            sums.append([float(np.abs(p[b]).sum()) for b in range(B)])
        out[tag] = {"deterministic": bool(sums[0] == sums[1] == sums[2]), "per_instance": [round(x, 6) for x in sums[0]], "run2": [round(x, 6) for x in sums[1]] if sums[0] != sums[1] else None}
    print(json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    import bench
    os.environ.setdefault("BTBA_BENCH_NPROC", "8")
    cfg = bench.CONFIGS["c3"]
    if not os.path.exists(CACHE):
        data = {"full": bench.generate_instances(cfg, list(range(8))), "masked": bench.generate_instances(cfg, list(range(8)), masked=True)}
        pickle.dump(data, open(CACHE, "wb"))
    for spec in sys.argv[1:]:
        lib, *sets = spec.split(":")
        env = dict(os.environ, BTBA_LIB_PATH=os.path.abspath(lib))
        env.update(dict(kv.split("=", 1) for kv in sets))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], env=env, capture_output=True, text=True, timeout=300)
        print(r.stdout.strip() if r.stdout.strip() else ("FAILED " + lib + " " + r.stderr[-600:]), flush=True)


if __name__ == "__main__":
    main()
