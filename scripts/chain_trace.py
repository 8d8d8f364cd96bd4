#!/usr/bin/env python
"""Workgroup timeline of ONE chained launch (btba_kernels.hpp: k_chain).  Run on the GPU box:

    BTBA_CHAIN_TRACE_FILE=gpurun_out/chain.bin python scripts/chain_trace.py [--masked] [--instances 32] [--out gpurun_out/chain_trace.json]

Every block of the launch stamps (start, end of its wait, end) in 100 MHz ticks plus (kind, iteration, instance, hardware id); this script
runs the bench workload a few times (the library dumps the timeline of every chained solve; the last one is kept) and summarises: how long the
launch is, how busy its 1 536 slots are, what the items of each kind cost, how long they wait, and -- the point of the design -- how much slack
every system solve has before its instance's next sweep items come up."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--masked", action="store_true")
    ap.add_argument("--instances", type=int, default=32)
    ap.add_argument("--out", default=None)
    ap.add_argument("--file", default=os.environ.get("BTBA_CHAIN_TRACE_FILE"))
    ap.add_argument("--analyse-only", action="store_true")
    args = ap.parse_args()
    if not args.analyse_only:
        import torch
        import bench
        from bundletrack_amd import _lib
        from bundletrack_amd.optimizer import BatchSolver, Workspace
        cfg = bench.CONFIGS["c3"]
        B, K = args.instances, cfg["K"]
        inst = bench.generate_instances(cfg, list(range(B)), args.masked)
        dev = torch.device("cuda:0")
        ws = Workspace()
        bs = BatchSolver(ws)
        if args.masked:
            bs.params.flags |= _lib.FLAG_COMPACTION
        corr, offs, mx = bs.pack_correspondences([p["corr"] for p in inst], K)
        zn_d = torch.from_numpy(np.stack([p["zn"] for p in inst])).to(dev)
        corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev)
        offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
        poses0 = torch.from_numpy(np.stack([p["poses"] for p in inst])).to(dev)
        aux = bs.cache_aux(zn_d, valid_lists=args.masked)
        use_c24 = not args.masked
        if use_c24:
            aux["corr24"] = bs.pack_correspondences24(corr_d, offs_d, mx, K)
        for _ in range(4):
            p = poses0.clone()
            bs.solve_zn(zn_d, inst[0]["H"], inst[0]["W"], inst[0]["K"], None if use_c24 else corr_d, offs_d, mx, p, aux=aux, corr_stride=corr_d.shape[1])
            ws.sync()
    raw = np.fromfile(args.file, dtype=np.uint64)
    nrec = int(raw[0])
    t = raw[1:1 + 4 * nrec].reshape(-1, 4)
    stamps = raw[1 + 4 * nrec:].reshape(-1, 8).astype(np.int64)
    t = t[t[:, 2] > 0]                                  # blocks that ran an item (unused slots of uneven batches stay zero)
    start, wait, end, tag = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2].astype(np.int64), t[:, 3]
    kind, it, b = (tag & np.uint64(0xFF)).astype(int), ((tag >> np.uint64(8)) & np.uint64(0xFF)).astype(int), ((tag >> np.uint64(16)) & np.uint64(0xFFFF)).astype(int)
    t0 = start.min()
    us = lambda x: (np.asarray(x, np.float64)) / 100.0
    span = float(us(end.max() - t0))
    res = {"blocks": int(len(t)), "launch_span_us": round(span, 1), "iterations": int(it.max() + 1), "instances": int(b.max() + 1)}
    busy = float(us((end - start).sum()))
    res["mean_resident_workgroups"] = round(busy / span, 1)
    for k, name in ((0, "dense"), (1, "sparse"), (2, "solve")):
        m = kind == k
        if not m.any():
            continue
        d, w = us(end[m] - wait[m]), us(wait[m] - start[m])
        res[name] = {"items": int(m.sum()), "work_us_mean": round(float(d.mean()), 2), "work_us_p90": round(float(np.percentile(d, 90)), 2), "work_us_max": round(float(d.max()), 2),
                     "wait_us_mean": round(float(w.mean()), 3), "wait_us_p99": round(float(np.percentile(w, 99)), 2), "wait_us_max": round(float(w.max()), 2),
                     "slot_us_total": round(float(us((end[m] - start[m]).sum())), 0), "wait_us_total": round(float(w.sum()), 0)}
    # phases of the solve items (system_solve_body's stamps: 6 start, 7 partials reduced, 0 tables staged, 2 system assembled, 3 PCG done, 4 poses updated)
    st = stamps[(stamps[:, 4] > 0) & (stamps[:, 6] > 0)]
    if len(st):
        ph = lambda a, c: round(float(us(st[:, c] - st[:, a]).mean()), 2)
        res["solve_phases_us"] = {"reduce": ph(6, 7), "stage_and_zero_fill": ph(7, 0), "assemble": ph(0, 2), "pcg": ph(5, 3), "update": ph(3, 4), "stamped_total": ph(6, 4)}
    # slack of every hand-off: solve (i, b) ends -> first sweep item of (i + 1, b) starts
    slack = []
    sm = kind == 2
    for i in range(int(it.max())):
        for bb in np.unique(b):
            s_end = end[sm & (it == i) & (b == bb)]
            nxt = start[(kind < 2) & (it == i + 1) & (b == bb)]
            if len(s_end) and len(nxt):
                slack.append(float(us(nxt.min() - s_end[0])))
    slack = np.array(slack)
    res["handoff_slack_us"] = {"mean": round(float(slack.mean()), 1), "min": round(float(slack.min()), 1), "p10": round(float(np.percentile(slack, 10)), 1),
                               "negative_share": round(float((slack < 0).mean()), 3)}
    # per iteration: when the first item starts and the last solve ends
    res["iteration_windows_us"] = [[round(float(us(start[it == i].min() - t0)), 1), round(float(us(end[it == i].max() - t0)), 1)] for i in range(int(it.max()) + 1)]
    last = it == it.max()
    res["tail_us_after_last_sweep_item"] = round(float(us(end.max() - end[last & (kind < 2)].max())), 1)
    # occupancy profile: resident (and waiting) workgroups sampled every 10 us
    grid = np.arange(0, span, 10.0)
    s_us, e_us, w_us = us(start - t0), us(end - t0), us(wait - t0)
    res["resident_every_10us"] = [int(((s_us <= g) & (e_us > g)).sum()) for g in grid]
    res["waiting_every_10us"] = [int(((s_us <= g) & (w_us > g)).sum()) for g in grid]
    print(json.dumps(res))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
