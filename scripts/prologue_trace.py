"""Where the dense workgroup's prologue goes (developer tool; needs a library built with -DBTBA_WG_TRACE -DBTBA_PROLOGUE_TRACE from the sources with scripts/dev/prologue_trace.patch applied: two more stamps).
    BTBA_LIB_PATH=build/ab/protrace.so python scripts/prologue_trace.py"""
import json, os, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from bundletrack_amd.optimizer import BatchSolver, Workspace
B = 32


def main():
    os.environ.setdefault("BTBA_BENCH_NPROC", "8")
    inst = bench.generate_instances(bench.CONFIGS["c3"], list(range(8)))
    pick = [inst[b % len(inst)] for b in range(B)]
    dev = torch.device("cuda:0"); ws = Workspace(); bs = BatchSolver(ws)
    corr, offs, mx = bs.pack_correspondences([p["corr"] for p in pick], 15)
    zn_d = torch.from_numpy(np.stack([p["zn"] for p in pick])).to(dev)
    corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
    poses0 = torch.from_numpy(np.stack([p["poses"] for p in pick])).to(dev)
    path = os.path.join(tempfile.gettempdir(), "pro_trace.bin")
    for rep in range(3):
        poses_d = poses0.clone()
        if rep == 2: os.environ["BTBA_WG_TRACE_FILE"] = path
        bs.solve_zn(zn_d, pick[0]["H"], pick[0]["W"], pick[0]["K"], corr_d, offs_d, mx, poses_d); ws.sync()
    q = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    kind = (q[:, 3] & np.uint64(0xFF)).astype(int); d = kind == 0
    a = (q[:, 1] & np.uint64(0xFFFFFF)).astype(np.float64) / 100.0; b = ((q[:, 1] >> np.uint64(24)) & np.uint64(0xFFFFFF)).astype(np.float64) / 100.0
    pro = ((q[:, 3] >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.float64) / 100.0
    def st(x): return {"mean": round(float(x.mean()), 2), "p10": round(float(np.percentile(x, 10)), 2), "p50": round(float(np.percentile(x, 50)), 2), "p90": round(float(np.percentile(x, 90)), 2)}
    print(json.dumps({"dense_items": int(d.sum()), "us_from_item_start": {"relative pose formed": st(a[d]), "tables written": st(b[d]), "live list built (loop starts)": st(pro[d])},
                      "phase_us": {"item decode + pose loads + 4x4 product": st(a[d]), "ray / coordinate tables": st((b - a)[d]), "hull tests + list compaction": st((pro - b)[d])}}))


if __name__ == "__main__":
    main()
