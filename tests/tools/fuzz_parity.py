"""Randomised parity sweep of the drop-in boundary against the CPU oracle (GPU box; developer tool, not collected by pytest).
    python tests/tools/fuzz_parity.py [n_cases] > gpurun_out/fuzz_parity.jsonl
Window sizes 2 ... 9, 0 ... 500 correspondences per pair with ragged pairs (some emptied, some thinned by invalidated entries),
object-masked and fully valid frames, different initial perturbations.  Prints one line per case: the worst pose difference of the
final iterate, rad and m.  A case above 1e-4 is not necessarily a failure (one flipped accept decision changes the trajectory:
tests/helpers.py explains a case by its decision trace); this tool only finds candidates."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from bundletrack_amd import synthetic as S
from bundletrack_amd.optimizer import BatchSolver, OptimizerGpu, Workspace
from oracle import oracle as O
from oracle import reference as R


def run_cases(n_cases, explain_always=False, only=None, hook=None, prepare=None):
    """Yields one record per case (see the module docstring).  only: solve just that case (the others still draw their random numbers);
    hook(rec, pb, corr, caches, ref, tv): called for explained cases with the problem and both traces; prepare(pb): may edit the problem before it is solved."""
    dev = torch.device("cuda:0")
    opt = OptimizerGpu()
    ws = Workspace()
    from fuzz_cases import fuzz_cases              # the generator is shared with the CPU-side tools (tests/fuzz_cases.py)
    for case, pb, corr, meta in fuzz_cases(n_cases, only=only):
        K, m, background, perturb = meta["K"], meta["corr_per_pair"], meta["background"], meta["perturb_deg"]
        if prepare is not None:                 # (experiments: e.g. replace the starting poses by a fixed point of Exp(Log(.)) -- after all random draws)
            prepare(pb)
        depths = [torch.from_numpy(pb.depth[k]).to(dev) for k in range(K)]
        normals = [torch.from_numpy(pb.normals[k]).to(dev) for k in range(K)]
        poses = pb.poses_init.copy()
        opt.optimizeFrames(corr, pb.n_match_per_pair, K, pb.H, pb.W, depths, None, normals, poses, pb.K)
        caches = [O.build_cache(pb.depth[k], pb.normals[k], pb.K) for k in range(K)]
        ref = O.solve(np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches]), caches[0]["intr"], corr, pb.poses_init)
        dr = max(S.pose_error(poses[k], ref.poses[k])[0] for k in range(K)); dt = max(S.pose_error(poses[k], ref.poses[k])[1] for k in range(K))
        rec = {"case": case, "K": K, "corr_per_pair": m, "background": background, "perturb_deg": perturb, "rot": float(dr), "trans": float(dt),
               "finite": bool(np.isfinite(poses).all())}
        # THE REFERENCE'S OWN SOLVER on the same inputs (oracle/_ref: solveBundlingStub through the launch emulator, forward order, IEEE), every iterate
        campos, nrm = np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches])
        have_ref = os.path.exists(R.SO_SOLVER)
        if have_ref:
            Pref, _, Tref = R.solve(campos, nrm, caches[0]["intr"], corr, pb.poses_init, want_iterates=True)
            rec["vs_reference_final"] = float(f"{max(max(S.pose_error(poses[k], Pref[k])) for k in range(K)):.3g}")
        if explain_always or max(dr, dt) >= 1e-4 or (have_ref and rec["vs_reference_final"] >= 1e-4):
            # explain it (tests/helpers.py): per-iterate traces of both sides, the first differing decision, the oracle's own summation-order spread
            from helpers import first_decision_divergence
            seq = O.solve(campos, nrm, caches[0]["intr"], corr, pb.poses_init, params=O.default_params(accum_mode=0))
            bs = BatchSolver(ws)
            cpk, offs, mx = bs.pack_correspondences([corr], K)
            corr_d = torch.from_numpy(cpk.view(np.uint8).reshape(1, -1, 32)).to(dev); offs_d = torch.from_numpy(offs.astype(np.int32)).to(dev)
            poses_d = torch.from_numpy(pb.poses_init[None].astype(np.float32)).to(dev)
            tv = bs.trace_view(bs.solve(torch.from_numpy(campos[None]).to(dev), torch.from_numpy(nrm[None]).to(dev), caches[0]["intr"], corr_d, offs_d, mx, poses_d, trace=True))
            div = first_decision_divergence(tv.pcg_scalars[0], tv.dense_pair[0][..., 27], ref.pcg_scalars, ref.dense_count)
            G = ref.T_after.shape[0]
            per_it = [max(max(S.pose_error(tv.T_after[0, it, k], ref.T_after[it, k])) for k in range(K)) for it in range(G)]
            spread = [max(max(S.pose_error(seq.T_after[it, k], ref.T_after[it, k])) for k in range(K)) for it in range(G)]
            first = div[0] if div is not None else G
            unexplained = [it for it in range(G) if it < first and per_it[it] >= max(1e-4, 3.0 * max(spread[:it + 1]))]
            if have_ref:
                # Round 6: THE LICENCE.  How far is the reference from ITSELF on this window -- its own code under three other legal execution orders of its float
                # atomics and under the model of its own -use_fast_math build (oracle/reference.py::self_spread, 7 more emulated solves)?  An iterate of the HIP
                # path above 1e-4 is `beyond_reference_spread` unless it is within 3x that spread (cumulative over the iterates so far).
                hip_ref = [max(max(S.pose_error(tv.T_after[0, it, k], Tref[it, k])) for k in range(K)) for it in range(G)]
                rec["vs_reference_per_iterate"] = [float(f"{x:.3g}") for x in hip_ref]
                if max(hip_ref) >= 1e-4 or rec["vs_reference_final"] >= 1e-4:
                    sp, runs = R.self_spread(campos, nrm, caches[0]["intr"], corr, pb.poses_init, S.pose_error)
                    cum = np.maximum.accumulate(sp)
                    nearest = [min(max(max(S.pose_error(tv.T_after[0, it, k], runs[v, it, k])) for k in range(K)) for v in range(runs.shape[0])) for it in range(G)]
                    # the drop-in boundary's result (compact cache, the library's own tile / chunk choice: other sums than the traced batch entry) has a final iterate only
                    b_near = min(max(max(S.pose_error(poses[k], runs[v, -1, k])) for k in range(K)) for v in range(runs.shape[0]))
                    def beyond_any(cum_):
                        return any(hip_ref[it] >= max(1e-4, 3.0 * cum_[it]) for it in range(G)) or rec["vs_reference_final"] >= max(1e-4, 3.0 * cum_[-1])
                    if beyond_any(cum):
                        # Eight runs SAMPLE what the reference does to itself; one flipped accept decision is a rare, large event (fuzz case 113: a dense-only four-frame
                        # window -- seven of the eight runs within 1e-4, the boundary path 1.2e-3 away; of 48 runs one lands 1.24e-3 from the forward run as well).
                        # Before a window is called beyond the reference's spread, 48 more runs (24 shuffles x IEEE / fast-math) are drawn.
                        more = [(f"shuffle:{sd}", fm) for sd in range(3, 27) for fm in (False, True)]
                        sp2, runs2 = R.self_spread(campos, nrm, caches[0]["intr"], corr, pb.poses_init, S.pose_error, variants=[("forward", False)] + more)
                        sp = np.maximum(sp, sp2); cum = np.maximum.accumulate(sp); runs = np.concatenate([runs, runs2[1:]])
                        nearest = [min(max(max(S.pose_error(tv.T_after[0, it, k], runs[v, it, k])) for k in range(K)) for v in range(runs.shape[0])) for it in range(G)]
                        b_near = min(max(max(S.pose_error(poses[k], runs[v, -1, k])) for k in range(K)) for v in range(runs.shape[0]))
                        rec["reference_runs"] = int(runs.shape[0])
                    rec.update({"reference_self_spread": [float(f"{x:.3g}") for x in sp], "hip_to_nearest_reference_run": [float(f"{x:.3g}") for x in nearest],
                                "beyond_reference_spread": [it for it in range(G) if hip_ref[it] >= max(1e-4, 3.0 * cum[it])],
                                "boundary_final_beyond_reference_spread": bool(rec["vs_reference_final"] >= max(1e-4, 3.0 * cum[-1])),
                                "boundary_final_to_nearest_reference_run": float(f"{b_near:.3g}"),
                                "reference_holds_1e-4_against_itself": bool(sp.max() < 1e-4)})
                    # the reference is the authority: an iterate the oracle-based rule cannot explain (its floor, the oracle's OpenMP summation spread, changes with the
                    # box's thread count) stands only if it is outside the reference's own spread too
                    unexplained = [it for it in unexplained if hip_ref[it] >= max(1e-4, 3.0 * cum[it])]
                else:
                    rec.update({"reference_self_spread": None, "beyond_reference_spread": [], "boundary_final_beyond_reference_spread": False})
                    unexplained = []                  # every iterate within 1e-4 of the reference's own forward run
            if unexplained:
                # Second opinion before calling an iterate unexplained: the oracle's sequential-sum run (`seq`) is an OpenMP reduction whose order -- and
                # with it `spread` -- changes with the box's thread count (case 13, an ill-conditioned K = 5 window with 5 matches per pair: 4.6e-5,
                # 9.9e-5 and 1.1e-4 at iterate 3 on three boxes, round 5).  Ground truth instead: the fp64 restatement of the same iterates
                # (oracle/oracle_np.py).  The HIP path may be at most 3x as far from it as the reference's own fp32 arithmetic (the oracle) is.
                from oracle import oracle_np as ON
                r64 = ON.solve(campos.astype(np.float64), nrm.astype(np.float64), caches[0]["intr"], corr, pb.poses_init.astype(np.float64))
                T64 = np.asarray(r64["T_after"])
                o64 = [max(max(S.pose_error(ref.T_after[it, k], T64[it, k])) for k in range(K)) for it in range(G)]
                h64 = [max(max(S.pose_error(tv.T_after[0, it, k], T64[it, k])) for k in range(K)) for it in range(G)]
                unexplained = [it for it in unexplained if h64[it] >= max(1e-4, 3.0 * max(o64[:it + 1]))]
                rec.update({"second_opinion_fp64": {"hip": [float(f"{x:.3g}") for x in h64], "oracle": [float(f"{x:.3g}") for x in o64]}})
            if hook: hook(rec, pb, corr, caches, ref, tv)
            rec.update({"first_divergence": None if div is None else [int(div[0]), div[1]], "per_iterate": [float(f"{x:.3g}") for x in per_it],
                        "oracle_own_spread": [float(f"{x:.3g}") for x in spread], "unexplained_iterates": unexplained})
        yield rec


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    only = int(sys.argv[2]) if len(sys.argv) > 2 else None
    worst = 0.0

    def against_fp64(rec, pb, corr, caches, ref, tv):
        # how far is EITHER fp32 evaluation from the fp64 restatement of the same iterate?  (oracle/oracle_np.py)
        from oracle import oracle_np as ON
        campos, nrm = np.stack([c["campos"] for c in caches]), np.stack([c["normals"] for c in caches])
        r64 = ON.solve(campos.astype(np.float64), nrm.astype(np.float64), caches[0]["intr"], corr, pb.poses_init.astype(np.float64))
        T64 = np.asarray(r64["T_after"])
        K = pb.n_frames
        rec["oracle32_vs_fp64"] = [float(f"{max(max(S.pose_error(ref.T_after[it, k], T64[it, k])) for k in range(K)):.3g}") for it in range(T64.shape[0])]
        # iterate 0 (identical inputs): right-hand sides and PCG scalars side by side, conditioning of the system the HIP path assembled
        A_h = tv.A[0, 0].astype(np.float64)[6:, 6:]               # frames 1 ..: [trans, rot] per frame
        b_h, b_o = tv.rhs[0, 0].astype(np.float64), np.asarray(ref.rhs[0], np.float64)
        w = np.linalg.eigvalsh(A_h)
        rec["iterate0"] = {"rhs_rel_diff": float(np.abs(b_h - b_o).max() / max(np.abs(b_o).max(), 1e-30)), "eig_min": float(w.min()), "eig_max": float(w.max()),
                           "cond_A_jacobi": float(np.linalg.cond(A_h / np.sqrt(np.outer(np.diag(A_h), np.diag(A_h))))),
                           "pcg_hip": [[float(f"{x:.6g}") for x in r] for r in tv.pcg_scalars[0, 0]], "pcg_oracle": [[float(f"{x:.6g}") for x in r] for r in np.asarray(ref.pcg_scalars[0])]}
        rec["hip_vs_fp64"] = [float(f"{max(max(S.pose_error(tv.T_after[0, it, k], T64[it, k])) for k in range(K)):.3g}") for it in range(T64.shape[0])]

    recs = []
    for rec in run_cases(n_cases, only=only, hook=against_fp64 if only is not None else None):
        worst = max(worst, rec["rot"], rec["trans"])
        recs.append(rec)
        print(json.dumps(rec), flush=True)
    above_ref = [r for r in recs if r.get("reference_self_spread")]      # a traced iterate or the boundary's final pose above 1e-4 against the reference's forward / IEEE run
    summary = {"cases": n_cases, "worst_vs_oracle": worst,
               "final_above_1e-4_vs_oracle": sum(max(r["rot"], r["trans"]) >= 1e-4 for r in recs),
               "final_above_1e-4_vs_reference": sum(r.get("vs_reference_final", 0.0) >= 1e-4 for r in recs),
               "above_1e-4_vs_reference (a traced iterate or the boundary's final pose)": len(above_ref),
               "of_those_the_reference_cannot_hold_to_1e-4_against_itself": sum(not r["reference_holds_1e-4_against_itself"] for r in above_ref),
               "windows_with_an_iterate_beyond_3x_the_reference_self_spread": [r["case"] for r in above_ref if r["beyond_reference_spread"]],
               "windows_whose_boundary_result_is_beyond_3x_the_reference_self_spread": [r["case"] for r in above_ref if r["boundary_final_beyond_reference_spread"]],
               "windows_above_1e-4_that_the_reference_itself_holds": [r["case"] for r in above_ref if r["reference_holds_1e-4_against_itself"]],
               "unexplained_by_the_decision_rule": [r["case"] for r in recs if r.get("unexplained_iterates")]}
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
