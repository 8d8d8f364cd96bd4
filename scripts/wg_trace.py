"""Workgroup timeline of one k_fused_sweeps launch (developer tool; needs a library built with -DBTBA_WG_TRACE, scripts/ab_build.sh).
    BTBA_LIB_PATH=build/ab/trace.so python scripts/wg_trace.py [--tiles T] > gpurun_out/wg_trace.json
Every workgroup records (start, end) in 100 MHz ticks and the CU it ran on; the summary answers: how long are dense / sparse
workgroups, how much of the launch is the drain at the end (slots idle while the last workgroups finish), how even is the load."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch


def main():
    import bench
    from bundletrack_amd import _lib
    from bundletrack_amd.optimizer import BatchSolver, Workspace
    tiles = int(sys.argv[sys.argv.index("--tiles") + 1]) if "--tiles" in sys.argv else 0
    B = 32
    os.environ.setdefault("BTBA_BENCH_NPROC", "8")
    inst = bench.generate_instances(bench.CONFIGS["c3"], list(range(8)))
    pick = [inst[b % len(inst)] for b in range(B)]
    dev = torch.device("cuda:0")
    ws = Workspace()
    bs = BatchSolver(ws)
    bs.params.dense_tiles = tiles
    corr, offs, mx = bs.pack_correspondences([p["corr"] for p in pick], 15)
    zn_d = torch.from_numpy(np.stack([p["zn"] for p in pick])).to(dev)
    corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
    poses0 = torch.from_numpy(np.stack([p["poses"] for p in pick])).to(dev)
    path = os.path.join(tempfile.gettempdir(), "wg_trace.bin")
    for rep in range(3):
        poses_d = poses0.clone()
        if rep == 2: os.environ["BTBA_WG_TRACE_FILE"] = path
        bs.solve_zn(zn_d, pick[0]["H"], pick[0]["W"], pick[0]["K"], corr_d, offs_d, mx, poses_d)
        ws.sync()
    q = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    t0 = q[:, 0].astype(np.int64); t1 = q[:, 1].astype(np.int64); hw = q[:, 2]; kind = (q[:, 3] & np.uint64(0xFF)).astype(int)
    pro = ((q[:, 3] >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.int64) / 100.0          # dense items: end of prologue / end of pixel loop, us from start
    loop_end = ((q[:, 3] >> np.uint64(32)) & np.uint64(0xFFFFFF)).astype(np.int64) / 100.0
    n_live = ((hw >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
    ok = t1 > 0
    base = t0[ok].min()
    s = (t0 - base) / 100.0; e = (t1 - base) / 100.0          # microseconds
    dur = e - s
    cu = ((hw >> np.uint64(16)) & np.uint64(0xF)) * np.uint64(4096) + ((hw >> np.uint64(8)) & np.uint64(0xFF))        # (xcc, se / sh / cu bits of HW_ID)
    span = float(e[ok].max())
    out = {"workgroups": int(ok.sum()), "missing": int((~ok).sum()), "span_us": round(span, 2), "distinct_cus": int(len(np.unique(cu[ok])))}
    for k, name in ((0, "dense"), (1, "sparse")):
        d = dur[ok & (kind == k)]
        out[name] = {"n": int(d.size), "mean_us": round(float(d.mean()), 2), "p10": round(float(np.percentile(d, 10)), 2), "p50": round(float(np.percentile(d, 50)), 2),
                     "p90": round(float(np.percentile(d, 90)), 2), "max": round(float(d.max()), 2), "sum_ms": round(float(d.sum()) / 1e3, 3)}
    dn = ok & (kind == 0)
    ph = {"prologue_us": pro[dn], "pixel_loop_us": (loop_end - pro)[dn], "epilogue_us": (dur - loop_end)[dn]}
    out["dense_phases"] = {k: {"mean": round(float(v.mean()), 2), "p10": round(float(np.percentile(v, 10)), 2), "p90": round(float(np.percentile(v, 90)), 2)} for k, v in ph.items()}
    out["dense_live_blocks"] = {"mean": round(float(n_live[dn].mean()), 1), "us_per_live_block_per_wave_trip": round(float((loop_end - pro)[dn].sum() / max(n_live[dn].sum(), 1) * 4), 3)}
    # running workgroups over time
    edges = np.linspace(0, span, 41)
    running = [(int(((s <= t) & (e > t) & ok).sum())) for t in edges[:-1] + (edges[1] - edges[0]) / 2]
    out["running_workgroups_per_2.5pct_of_span"] = running
    out["mean_running"] = round(float(dur[ok].sum() / span), 1)
    out["last_start_us"] = round(float(s[ok].max()), 2)
    per_cu_end = {}
    for c, ee in zip(cu[ok], e[ok]): per_cu_end[c] = max(per_cu_end.get(c, 0.0), ee)
    ends = np.array(list(per_cu_end.values()))
    out["cu_finish_us"] = {"min": round(float(ends.min()), 2), "p10": round(float(np.percentile(ends, 10)), 2), "p50": round(float(np.percentile(ends, 50)), 2), "max": round(float(ends.max()), 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
