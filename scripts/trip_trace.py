"""Where a wave's time goes inside one block trip of the fused sweep (developer tool; needs a library built with
-DBTBA_WG_TRACE -DBTBA_TRIP_TRACE).
    BTBA_LIB_PATH=build/ab/triptrace.so python scripts/trip_trace.py > gpurun_out/trip_trace.json
Wave 0 of every dense workgroup sums, in shader-clock cycles (s_memtime), over its trips:
    top    from the start of a trip to the validity ballot: wait for the block's pixels (prefetched), ray tables from LDS, projection
    taps   from there to the arrival of the four tap gathers (normal rotation, address arithmetic and the blend weights run meanwhile)
    tail   from there to the last accumulate: blend, accept tests, 28 accumulates
The stamps perturb the schedule a little (each is an s_memtime + wait)."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch


def main():
    import bench
    from bundletrack_amd.optimizer import BatchSolver, Workspace
    B = 32
    os.environ.setdefault("BTBA_BENCH_NPROC", "8")
    inst = bench.generate_instances(bench.CONFIGS["c3"], list(range(8)))
    pick = [inst[b % len(inst)] for b in range(B)]
    dev = torch.device("cuda:0")
    ws = Workspace()
    bs = BatchSolver(ws)
    bs.params.dense_tiles = int(os.environ.get("BTBA_BENCH_TILES", "0"))          # 0 = the library's choice
    corr, offs, mx = bs.pack_correspondences([p["corr"] for p in pick], 15)
    zn_d = torch.from_numpy(np.stack([p["zn"] for p in pick])).to(dev)
    corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
    poses0 = torch.from_numpy(np.stack([p["poses"] for p in pick])).to(dev)
    path = os.path.join(tempfile.gettempdir(), "trip_trace.bin")
    for rep in range(3):
        poses_d = poses0.clone()
        if rep == 2: os.environ["BTBA_WG_TRACE_FILE"] = path
        bs.solve_zn(zn_d, pick[0]["H"], pick[0]["W"], pick[0]["K"], corr_d, offs_d, mx, poses_d)
        ws.sync()
    q = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    kind = (q[:, 3] & np.uint64(0xFF)).astype(int)
    live = ((q[:, 2] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
    dead = ((q[:, 2] >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.int64)      # (tt_dead << 32) of wg_dbg[7], shifted by another 32 -> only its low 16 bits survive: enough
    top = q[:, 0].astype(np.float64); taps = q[:, 1].astype(np.float64); tail = (q[:, 3] >> np.uint64(8)).astype(np.float64)
    d = (kind == 0) & (live + dead > 0)
    trips = live + dead
    out = {"dense_workgroups": int(d.sum()), "wave0_trips": int(trips[d].sum()), "live_trips": int(live[d].sum()), "dead_trips": int(dead[d].sum()),
           "cycles_per_trip": {"top (all trips)": round(float(top[d].sum() / trips[d].sum()), 1),
                               "taps in flight (live trips)": round(float(taps[d].sum() / max(live[d].sum(), 1)), 1),
                               "blend + accumulate (live trips)": round(float(tail[d].sum() / max(live[d].sum(), 1)), 1)}}
    lv = d & (live > 0)
    for name, v, n in (("top", top, trips), ("taps", taps, live), ("tail", tail, live)):
        x = v[lv] / np.maximum(n[lv], 1)
        out["per_workgroup_" + name] = {"p10": round(float(np.percentile(x, 10)), 1), "p50": round(float(np.percentile(x, 50)), 1), "p90": round(float(np.percentile(x, 90)), 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
