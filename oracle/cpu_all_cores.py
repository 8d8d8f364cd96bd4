"""Worker process of bench.py's all-core CPU baseline (test infrastructure, like everything under oracle/): loads instances from
an .npz, runs n_jobs whole solves on n_workers OpenMP threads (one single-threaded solve per thread, oracle.solve_jobs) and prints
one JSON line.  A separate process so that the bench can bound it with a timeout and so that nothing of the GPU process (HIP
runtime threads, torch's OpenMP pool) shares its thread pool.
    python oracle/cpu_all_cores.py <instances.npz> <max_workers> <weight_dense> <budget_seconds>"""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def cpu_quota():
    """CPUs the container may actually use: cgroup v2 cpu.max (quota / period) or v1 cfs_quota, else None (unlimited)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def main():
    from oracle import oracle as O
    path, n_workers, wd, budget = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
    z = np.load(path)
    B = int(z["n"])
    insts = [dict(campos=z[f"campos{b}"], normals=z[f"normals{b}"], intr=z[f"intr{b}"], corr=z[f"corr{b}"].view(O.ENTRYJ_DTYPE).reshape(-1), poses=z[f"poses{b}"]) for b in range(B)]
    prm = O.default_params(weight_dense_depth=wd)
    t_start = time.perf_counter()
    O.solve_jobs(insts, 8, 8, prm)                     # warm the thread pool and the page cache
    # more workers as long as it pays and the budget lasts: two solves per worker per stage, so a stage costs about two solve times
    # when the CPUs are really there (a container with a CPU quota below its affinity mask stops scaling early)
    stages, best = [], None
    quota = cpu_quota()
    if quota is not None:
        n_workers = max(1, min(n_workers, int(quota + 0.999)))      # threads beyond the container's CPU quota only time-share
    w = 16
    ladder = []
    while w < n_workers:
        ladder.append(w); w *= 2
    ladder.append(n_workers)
    for w in ladder:
        t0 = time.perf_counter()
        done = O.solve_jobs(insts, 2 * w, w, prm)
        dt = time.perf_counter() - t0
        st = {"workers": w, "solves": done, "seconds": round(dt, 3), "gn_iters_per_s": round(7.0 * done / dt, 2)}
        stages.append(st)
        if best is None or st["gn_iters_per_s"] > best["gn_iters_per_s"]:
            best = st
        print(json.dumps({"stage": st}), flush=True)
        if time.perf_counter() - t_start > budget or st["gn_iters_per_s"] < 0.8 * best["gn_iters_per_s"]:
            break
    print(json.dumps({"best": best, "stages": stages, "instances": B, "cpu_quota": quota}), flush=True)


if __name__ == "__main__":
    main()
