"""Worker process of bench.py's all-core CPU baseline (test infrastructure, like everything under oracle/): loads instances from
an .npz, runs n_jobs whole solves on n_workers OpenMP threads (one single-threaded solve per thread, oracle.solve_jobs) and prints
one JSON line.  A separate process so that the bench can bound it with a timeout and so that nothing of the GPU process (HIP
runtime threads, torch's OpenMP pool) shares its thread pool.
    python oracle/cpu_all_cores.py <instances.npz> <n_workers> <n_jobs> <weight_dense>"""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    from oracle import oracle as O
    path, n_workers, n_jobs, wd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    z = np.load(path)
    B = int(z["n"])
    insts = [dict(campos=z[f"campos{b}"], normals=z[f"normals{b}"], intr=z[f"intr{b}"], corr=z[f"corr{b}"].view(O.ENTRYJ_DTYPE).reshape(-1), poses=z[f"poses{b}"]) for b in range(B)]
    prm = O.default_params(weight_dense_depth=wd)
    O.solve_jobs(insts, min(n_workers, 8), min(n_workers, 8), prm)          # warm the thread pool and the page cache
    t0 = time.perf_counter()
    done = O.solve_jobs(insts, n_jobs, n_workers, prm)
    dt = time.perf_counter() - t0
    print(json.dumps({"solves": done, "seconds": dt, "workers": n_workers, "instances": B}), flush=True)


if __name__ == "__main__":
    main()
