import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from bundletrack_amd.optimizer import BatchSolver, Workspace


def main():
    os.environ["BTBA_BENCH_NPROC"] = "1"
    inst = bench.generate_instances(bench.CONFIGS["c3"], [0])
    dev = torch.device("cuda:0"); ws = Workspace()
    for tiles, chunks in ((0, 0), (4, 0), (5, 0), (8, 0), (15, 0), (0, 2), (0, 4), (0, 8), (5, 4), (4, 4)):
        bs = BatchSolver(ws); bs.params.dense_tiles = tiles; bs.params.sparse_chunks = chunks
        p = inst[0]
        corr, offs, mx = bs.pack_correspondences([p["corr"]], 15)
        zn_d = torch.from_numpy(p["zn"][None]).to(dev)
        corr_d = torch.from_numpy(corr.view(np.uint8).reshape(1, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
        poses0 = torch.from_numpy(p["poses"][None]).to(dev); poses_d = poses0.clone()
        def step():
            poses_d.copy_(poses0); bs.solve_zn(zn_d, p["H"], p["W"], p["K"], corr_d, offs_d, mx, poses_d)
        for _ in range(20): step()
        ws.sync(); st0 = ws.collect_stats()
        t0 = time.perf_counter()
        for _ in range(300): step()
        ws.sync(); dt = (time.perf_counter() - t0) / 300
        st = ws.collect_stats()
        print(json.dumps({"tiles_asked": tiles, "chunks_asked": chunks, "tiles": st["dense_tiles"], "chunks": st["sparse_chunks"], "ms_per_solve": round(dt * 1e3, 4),
                          "checksum": float(np.abs(poses_d.cpu().numpy()).sum())}), flush=True)


if __name__ == "__main__":
    main()
