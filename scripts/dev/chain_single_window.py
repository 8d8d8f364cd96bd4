"""Is the chained launch (BTBA_OPT_CHAIN = 1: all seven iterations in ONE launch) worth it for ONE window, where the 14 launch gaps of the plain schedule are a quarter of the device time?
    python scripts/dev/chain_single_window.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    os.environ["BTBA_BENCH_NPROC"] = "1"
    import numpy as np, torch, bench, ab_solve
    from bundletrack_amd import _lib
    from bundletrack_amd.optimizer import Workspace
    dev = torch.device("cuda:0")
    cfg = bench.CONFIGS["c3"]
    for masked in (True, False):
        inst = bench.generate_instances(cfg, [0], masked)
        for chain, tiles, chunks in ((0, 0, 0), (0, 2, 2), (1, 2, 2), (1, 1, 2), (1, 2, 1), (1, 1, 1)):
            ws = Workspace()
            ws.set_option(_lib.OPT_CHAIN, chain)
            bs, step, poses_d = ab_solve.setup(ws, cfg, inst, 1, masked, dev)
            bs.params.flags &= ~(_lib.FLAG_TIME_KERNELS | _lib.FLAG_TIME_SAMPLED)
            bs.params.dense_tiles = tiles; bs.params.sparse_chunks = chunks
            for _ in range(10): step()
            ws.sync()
            t0 = time.perf_counter()
            n = 200
            for _ in range(n): step()
            ws.sync()
            dt = (time.perf_counter() - t0) / n
            st = ws.collect_stats()
            print(json.dumps({"masked": masked, "chain": chain, "tiles": st["dense_tiles"], "chunks": st["sparse_chunks"], "chained_iterations": st["chain_iterations"], "ms_per_solve": round(dt * 1e3, 4),
                              "checksum": float(np.abs(poses_d.cpu().numpy()).sum())}), flush=True)
            ws.close()


if __name__ == "__main__":
    main()
