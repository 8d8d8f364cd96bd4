"""Tiles per dense pair / chunks per correspondence segment for ONE window (the tracker's call), round 5's solve kernel: ms per solve, no event brackets.
    python scripts/dev/single_window_tiles.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    os.environ["BTBA_BENCH_NPROC"] = "1"
    import numpy as np, torch, bench, ab_solve
    from bundletrack_amd import _lib
    from bundletrack_amd.optimizer import Workspace
    dev = torch.device("cuda:0")
    ws = Workspace()
    for name, K, m in (("c3", 15, 2000), ("K10", 10, 1000), ("K5", 5, 300)):
        cfg = dict(bench.CONFIGS["c3"], K=K, m=m)
        for masked in (True, False):
            inst = bench.generate_instances(cfg, [0], masked)
            for B in (1, 2):
                res = {}
                combos = [(0, 0)] + [(t, c) for t in ((1, 2, 3, 4) if masked else (2, 4, 6, 8, 10)) for c in (1, 2, 4)]
                for tiles, chunks in combos:
                    bs, step, poses_d = ab_solve.setup(ws, cfg, inst, B, masked, dev)
                    bs.params.flags &= ~(_lib.FLAG_TIME_KERNELS | _lib.FLAG_TIME_SAMPLED)
                    bs.params.dense_tiles = tiles; bs.params.sparse_chunks = chunks
                    for _ in range(8): step()
                    ws.sync()
                    best = 1e9
                    for rep in range(3):
                        t0 = time.perf_counter()
                        for _ in range(100): step()
                        ws.sync()
                        best = min(best, (time.perf_counter() - t0) / 100)
                    st = ws.collect_stats()
                    res[f"{st['dense_tiles']}x{st['sparse_chunks']}" + ("*" if tiles == 0 else "")] = round(best * 1e3, 4)
                print(json.dumps({"window": name, "masked": masked, "B": B, "ms_per_solve by tiles x chunks (* = the library's choice)": res}), flush=True)


if __name__ == "__main__":
    main()
