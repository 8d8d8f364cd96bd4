"""Where a PCG step of k_solve_small goes (developer build -DBTBA_SOLVE_PCG_STAMPS: the trace's eight clock slots = seven points inside step 1, wave 0's view).
    BTBA_LIB_PATH=build/ab/pcg_stamps.so python scripts/dev/pcg_stamps.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    os.environ["BTBA_BENCH_NPROC"] = "1"
    import torch, bench, ab_solve
    from bundletrack_amd import _lib
    from bundletrack_amd.optimizer import Workspace
    dev = torch.device("cuda:0"); ws = Workspace()
    cfg = bench.CONFIGS["c3"]; inst = bench.generate_instances(cfg, [0, 1])
    for B in (1, 32):
        bs, step, poses_d = ab_solve.setup(ws, cfg, inst, B, False, dev)
        bs.params.flags &= ~(_lib.FLAG_TIME_KERNELS | _lib.FLAG_TIME_SAMPLED)
        tv = bs.trace_view(step(trace=True))
        c = tv.clk.mean(axis=(0, 1))
        print(json.dumps({"B": B, "matvec": float(c[1] - c[0]), "barrier1": float(c[2] - c[1]), "read_Ap+pAp_reduce": float(c[3] - c[2]), "alpha+update+rz_reduce": float(c[4] - c[3]),
                          "beta+p+write": float(c[5] - c[4]), "barrier2": float(c[6] - c[5]), "step": float(c[6] - c[0])}), flush=True)


if __name__ == "__main__":
    main()
