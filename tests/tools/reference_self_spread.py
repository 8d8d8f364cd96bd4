"""THE REFERENCE AGAINST ITSELF (round 6; CPU only -- no GPU, no product code is measured here).

    python tests/tools/reference_self_spread.py [--workers 7] [--sets fuzz,random40,session,configs] > profiles/r06/reference_self_spread.json

The reference sums with float atomicAdd everywhere (SolverBundlingDenseUtil.h:217-285, SolverBundling.cu:575-818) and is built with -use_fast_math
(CMakeLists.txt:7): its own results move from run to run and from build to build.  Rounds 1-5 ran its code in ONE order with IEEE arithmetic.  This tool
runs the reference's own solver (oracle/_ref, the launch emulator of oracle/ref_shim/cuda_runtime.h) on every window of the parity suite under

    execution orders   forward | reverse | shuffle:1 | shuffle:2     (a legal order of every launch's (block, thread) cells each)
    arithmetic         IEEE | the fast-math model (contraction, approximate division / square root, perturbed sin / cos, flush-to-zero)

and records, per window and Gauss-Newton iterate, how far the reference is from ITSELF: `order` = worst pose difference (max of rad, m) of the three
other orders from the forward run, IEEE arithmetic; `fastmath` = the same over the four fast-math runs; `all` = both.  tests/helpers.py uses these
spreads (computed on the spot for the windows that need them) as the licence for an iterate above 1e-4: max(1e-4, 3 x the reference's own spread)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
from bundletrack_amd import synthetic as S
from oracle import oracle as O
from oracle import reference as R


def spread_record(name, campos, normals, intr, corr, poses, wd=1.0, meta=None):
    t0 = time.time()
    N = len(poses)
    _, T = R.self_spread(campos, normals, intr, corr, poses, S.pose_error, weight_dense=wd)
    base = T[0]
    G = base.shape[0]

    def worst(rows):
        return [float(f"{max(max(max(S.pose_error(T[v, it, k], base[it, k])) for k in range(N)) for v in rows):.3g}") for it in range(G)]
    rec = {"window": name, "n_frames": int(N), "order": worst([1, 2, 3]), "fastmath": worst([4, 5, 6, 7]), "all": worst(range(1, 8)),
           "moved_by_last_iteration": float(f"{max(max(S.pose_error(base[-1, k], base[-2, k])) for k in range(N)):.3g}") if G > 1 else None,
           "seconds": round(time.time() - t0, 1)}
    if meta:
        rec.update(meta)
    return rec


def job_fuzz(case):
    from fuzz_cases import fuzz_cases
    for c, pb, corr, meta in fuzz_cases(case + 1, only=case):
        caches = [O.build_cache(pb.depth[k], pb.normals[k], pb.K) for k in range(pb.n_frames)]
        return spread_record(f"fuzz/{c}", np.stack([x["campos"] for x in caches]), np.stack([x["normals"] for x in caches]), caches[0]["intr"], corr, pb.poses_init, meta=meta)


def random40_windows():
    """The forty windows of tests/test_gpu_vs_reference.py::test_random_windows_match_the_reference_solver (same draws)."""
    rng = np.random.default_rng(2024)
    out = []
    for trial in range(40):
        K = int(rng.integers(2, 10)); m = int(rng.choice([0, 40, 150, 400])); bg = bool(rng.integers(0, 2)); wd = float(rng.choice([0.0, 1.0, 1.0]))
        if m == 0 and wd == 0.0:
            wd = 1.0
        out.append(dict(trial=trial, K=K, m=m, bg=bg, wd=wd, perturb_deg=float(rng.uniform(0.5, 3.0)), perturb_m=float(rng.uniform(0.001, 0.008))))
    return out


def job_random40(w):
    pb = S.make_problem(w["K"], w["m"], 5000 + w["trial"], background=w["bg"], full_res=False, perturb_deg=w["perturb_deg"], perturb_m=w["perturb_m"])
    campos, normals, intr = S.analytic_cache(pb)
    return spread_record(f"random40/{w['trial']}", campos, normals, intr, pb.corr, pb.poses_init, wd=w["wd"],
                         meta={"K": w["K"], "corr_per_pair": w["m"], "background": w["bg"], "w_dense": w["wd"]})


def job_config(name):
    cfg = {"c2": (10, 1000, 0.0, S.config_seed(2), True), "c3": (15, 2000, 1.0, S.config_seed(3), True), "c3-masked": (15, 2000, 1.0, S.config_seed(3), False),
           "c3-benched-instance-0": (15, 2000, 1.0, S.config_seed(5, 0), True), "window-K5": (5, 300, 1.0, 28, False)}
    if name == "c4":
        pb = S.make_problem(30, 4000, S.config_seed(4), background=True, full_res=False, angles=S.pruned_pool_angles(60, 30, S.config_seed(4)))
        wd = 1.0
    else:
        K, m, wd, seed, bg = cfg[name]
        pb = S.make_problem(K, m, seed, background=bg, full_res=False)
    campos, normals, intr = S.analytic_cache(pb)
    return spread_record(f"config/{name}", campos, normals, intr, pb.corr, pb.poses_init, wd=wd)


def session_calls(n_frames=60):
    """The c1 tracking session (tests/test_tracking_session.py) driven by the CPU oracle; returns every bundle-adjustment call's inputs."""
    from helpers import OracleOptimizer
    from test_tracking_session import run_session
    calls = []

    class Recording(OracleOptimizer):
        def optimizeFrames(self, global_corres, n_match_per_pair, n_frames, H, W, depths, colors, normals, poses, K):
            caches = [self.O.build_cache(self._np(depths[k]).reshape(H, W), self._np(normals[k]).reshape(H, W, 4), K) for k in range(n_frames)]
            calls.append((np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches]), caches[0]["intr"], np.array(global_corres, copy=True),
                          np.array(poses, np.float32, copy=True)))
            return super().optimizeFrames(global_corres, n_match_per_pair, n_frames, H, W, depths, colors, normals, poses, K)
    run_session(Recording(O), n_frames)
    return calls


_SESSION = None


def job_session(i):
    global _SESSION
    if _SESSION is None:
        _SESSION = session_calls()
    campos, normals, intr, corr, poses = _SESSION[i]
    return spread_record(f"session/{i}", campos, normals, intr, corr, poses)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=7)
    ap.add_argument("--sets", default="configs,fuzz,random40,session")
    ap.add_argument("--fuzz", type=int, default=120)
    a = ap.parse_args()
    import multiprocessing as mp
    jobs = []
    sets = a.sets.split(",")
    if "configs" in sets:
        jobs += [(job_config, n) for n in ("c4", "c3", "c3-masked", "c3-benched-instance-0", "c2", "window-K5")]          # longest first
    if "fuzz" in sets:
        jobs += [(job_fuzz, c) for c in range(a.fuzz)]
    if "random40" in sets:
        jobs += [(job_random40, w) for w in random40_windows()]
    if "session" in sets:
        jobs += [(job_session, i) for i in range(59)]
    t0 = time.time()
    with mp.Pool(a.workers) as pool:
        res = [pool.apply_async(f, (x,)) for f, x in jobs]
        recs = []
        for r in res:
            rec = r.get()
            recs.append(rec)
            print(json.dumps(rec), file=sys.stderr, flush=True)
    finals = [r["all"][-1] for r in recs]
    worst_any = [max(r["all"]) for r in recs]
    summary = {"what": "reference vs reference: worst pose difference (max of rad, m) per Gauss-Newton iterate between the reference's own solver run forward / IEEE and "
                       "run under 3 other execution orders (`order`), under the fast-math model in 4 orders (`fastmath`), or any of the 7 (`all`)",
               "variants": [f"{o}{' fast-math' if fm else ''}" for o, fm in R.VARIANTS], "windows": len(recs),
               "final_iterate_above_1e-4": int(sum(f >= 1e-4 for f in finals)), "any_iterate_above_1e-4": int(sum(w >= 1e-4 for w in worst_any)),
               "final_iterate_above_1e-4_by_order_alone": int(sum(r["order"][-1] >= 1e-4 for r in recs)),
               "host_cpus": os.cpu_count(), "seconds": round(time.time() - t0, 1)}
    json.dump({"summary": summary, "windows": recs}, sys.stdout, indent=None)
    print()


if __name__ == "__main__":
    main()
