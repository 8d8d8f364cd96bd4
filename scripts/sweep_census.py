"""Lane census of the dense block walk (developer tool; needs a library built with -DBTBA_CENSUS: `scripts/ab_build.sh census -DBTBA_CENSUS` or
build/ab/census.so):   BTBA_LIB_PATH=build/ab/census.so python scripts/sweep_census.py [--config c3|c4] [--masked] > profiles/r06/sweep_census.json
Where do the lanes of a walked 8 x 8 block die?  Per solve (7 launches) of the bench batch: wave trips, lanes with a usable source depth, lanes whose
projection lands in the target image, trips that end at ballot(valid) == 0, lanes in the heavy half (taps, blend, 27-FMA accumulation), lanes accepted.
Decides whether compacting the survivors of two blocks before the accumulation can pay (round 5's verdict, item 4a/c)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch


def main():
    import bench
    from bundletrack_amd import _lib
    from bundletrack_amd.optimizer import BatchSolver, Workspace
    cfg_name = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "c3"
    cfg = bench.CONFIGS[cfg_name]
    B, K = 32, cfg["K"]
    os.environ.setdefault("BTBA_BENCH_NPROC", "8")
    inst = bench.generate_instances(cfg, list(range(8)))
    pick = [inst[b % len(inst)] for b in range(B)]
    dev = torch.device("cuda:0")
    ws = Workspace()
    bs = BatchSolver(ws)
    corr, offs, mx = bs.pack_correspondences([p["corr"] for p in pick], K)
    zn_d = torch.from_numpy(np.stack([p["zn"] for p in pick])).to(dev)
    corr_d = torch.from_numpy(corr.view(np.uint8).reshape(B, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
    poses0 = torch.from_numpy(np.stack([p["poses"] for p in pick])).to(dev)
    L = _lib.lib()
    L.btba_dev_census.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    poses_d = poses0.clone()
    bs.solve_zn(zn_d, pick[0]["H"], pick[0]["W"], pick[0]["K"], corr_d, offs_d, mx, poses_d)      # warm
    ws.sync()
    ws.set_option(_lib.OPT_COUNT_LIVE, 1)
    poses_d = poses0.clone()
    bs.solve_zn(zn_d, pick[0]["H"], pick[0]["W"], pick[0]["K"], corr_d, offs_d, mx, poses_d)
    ws.sync()
    out = (C.c_uint64 * 8)()
    assert L.btba_dev_census(ws._h, out) == 0
    v = [int(x) for x in out]
    npix = zn_d.shape[2] * zn_d.shape[3]
    pair_px = 7 * B * (K * (K - 1) // 2) * npix
    trips, src_ok, valid, dead_trips, heavy_lanes, accepted, rej_depth, rej_geom = v
    rec = {"workload": f"{cfg_name} x {B}, one solve (7 launches)", "pair_pixels": pair_px, "wave_trips": trips, "walked_lanes": 64 * trips,
           "share_of_pair_pixels_walked": round(64 * trips / pair_px, 4),
           "lanes_with_usable_source_depth": src_ok, "lanes_projecting_into_the_target (valid)": valid,
           "trips_ending_at_ballot_valid_0": dead_trips, "share_of_trips_ending_early": round(dead_trips / max(trips, 1), 4),
           "lanes_in_heavy_half": heavy_lanes, "lanes_accepted": accepted, "valid_but_target_depth_out_of_range": rej_depth, "valid_depth_ok_but_normal_or_distance_rejected": rej_geom,
           "heavy_half": {"share_alive_at_entry (valid / lanes)": round(valid / max(heavy_lanes, 1), 4), "share_accepted": round(accepted / max(heavy_lanes, 1), 4),
                          "dead_before_the_taps (not valid)": round(1 - valid / max(heavy_lanes, 1), 4), "dead_after_the_taps (valid, rejected)": round((valid - accepted) / max(heavy_lanes, 1), 4)},
           "accepted_share_of_pair_pixels": round(accepted / pair_px, 4)}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
