"""Test-side helpers.  Only tests may touch oracle/: the oracle-backed optimiser below has the signature of
OptimizerGpu.optimizeFrames so that bundler.Bundler can be driven by either backend."""
import numpy as np


class OracleOptimizer:
    """OptimizerGpu::optimizeFrames on the CPU oracle: full-resolution frames (numpy or torch) -> frame cache -> solve."""

    def __init__(self, oracle, **param_overrides):
        self.O = oracle
        self.params = oracle.default_params(**param_overrides)
        self.calls = []

    @staticmethod
    def _np(a):
        return a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)

    def optimizeFrames(self, global_corres, n_match_per_pair, n_frames, H, W, depths, colors, normals, poses, K):
        caches = [self.O.build_cache(self._np(depths[k]).reshape(H, W), self._np(normals[k]).reshape(H, W, 4), K) for k in range(n_frames)]
        campos = np.stack([c["campos"] for c in caches])
        nrm = np.stack([c["normals"] for c in caches])
        tr = self.O.solve(campos, nrm, caches[0]["intr"], global_corres, poses, params=self.params, want_trace=False)
        poses[...] = tr.poses.reshape(n_frames, 4, 4)
        return tr


class ParityOptimizer:
    """Runs the HIP optimiser and the oracle on identical inputs, records the worst pose difference of every call
    and hands the HIP result on (so a whole tracking session is driven by the product path)."""

    def __init__(self, hip_opt, oracle_opt, pose_error, classify=None, classify_above=1e-4):
        self.hip, self.ora, self.pose_error = hip_opt, oracle_opt, pose_error
        self.diffs = []
        self.classify = classify          # callable(call inputs) -> first differing decision, run on the calls above classify_above
        self.classify_above = classify_above
        self.divergences = {}

    def optimizeFrames(self, global_corres, n_match_per_pair, n_frames, H, W, depths, colors, normals, poses, K):
        ref = np.array(poses, np.float32, copy=True)
        self._remember(poses)
        self.ora.optimizeFrames(global_corres, n_match_per_pair, n_frames, H, W, depths, colors, normals, ref, K)
        self.hip.optimizeFrames(global_corres, n_match_per_pair, n_frames, H, W, depths, colors, normals, poses, K)
        self.diffs.append(max(max(self.pose_error(poses[k], ref[k])) for k in range(n_frames)))
        if self.classify and self.diffs[-1] >= self.classify_above:
            self.divergences[len(self.diffs) - 1] = self.classify(global_corres, n_frames, H, W, depths, normals, self._start, K)

    def _remember(self, poses):
        self._start = np.array(poses, np.float32, copy=True)


# ---- decision traces ("compare decisions first, then values") ---------------------------------
def first_decision_divergence(hip_pcg, hip_counts, ora_pcg, ora_counts):
    """First discrete decision the HIP path and the oracle take differently, or None.
    pcg [G, L, 4] = (pAp, alpha, r.z, beta) per Gauss-Newton iterate and PCG step; counts [G, Pd] accepted pixels per dense pair.
    Decisions: a dense pair accepting a different number of pixels (the accept tests <= 2 cm, >= cos 45 deg, depth range, in-image,
    SolverBundlingDenseUtil.h:78-110), alpha = 0 (pAp <= 1e-6) and beta = 0 (r.z <= 1e-6) of PCGStep_Kernel2/3
    (SolverBundling.cu:746-818).  Returns (iterate, kind, detail)."""
    G = ora_pcg.shape[0]
    for it in range(G):
        if hip_counts is not None and ora_counts is not None and hip_counts.shape[-1] and ora_counts.shape[-1]:
            hc, oc = np.rint(hip_counts[it]).astype(np.int64), np.asarray(ora_counts[it], np.int64)
            n = min(hc.shape[0], oc.shape[0])
            bad = np.nonzero(hc[:n] != oc[:n])[0]
            if bad.size:
                return (it, "accept", {"pair": int(bad[0]), "hip": int(hc[bad[0]]), "oracle": int(oc[bad[0]]), "pairs_differing": int(bad.size)})
        for li in range(ora_pcg.shape[1]):
            for col, kind in ((1, "alpha"), (3, "beta")):
                if (hip_pcg[it, li, col] == 0) != (ora_pcg[it, li, col] == 0):
                    return (it, kind, {"pcg_step": li, "hip": float(hip_pcg[it, li, col]), "oracle": float(ora_pcg[it, li, col]),
                                       "guarded_value_hip": float(hip_pcg[it, li, col - 1]), "guarded_value_oracle": float(ora_pcg[it, li, col - 1])})
    return None


def reference_licence(R, pose_error, campos, normals, intr, corr, poses, variants=None, **solve_kw):
    """THE LICENCE for an iterate above 1e-4 (round 6): how far is the reference from ITSELF on this window?  Runs the reference's own solver (oracle/_ref) under
    three other legal execution orders of its float atomics and under the model of its own -use_fast_math build (oracle/reference.py::self_spread) and returns
    (cum [G] = per iterate the largest pose difference of any of those runs from the forward / IEEE run, cumulative over the iterates so far;
     runs [V, G, N, 4, 4] = every run's iterates, runs[0] the forward / IEEE one)."""
    spread, runs = R.self_spread(campos, normals, intr, corr, poses, pose_error, variants=variants, **solve_kw)
    return np.maximum.accumulate(spread), runs


def beyond_reference_spread(hip_T, ref_T, cum_spread, pose_error, tol=1e-4, factor=3.0):
    """Iterates at which the HIP path is further from the reference's forward / IEEE run than max(tol, factor x the reference's own spread so far)."""
    G, N = ref_T.shape[0], ref_T.shape[1]
    err = [max(max(pose_error(hip_T[it, k], ref_T[it, k])) for k in range(N)) for it in range(G)]
    return [it for it in range(G) if err[it] >= max(tol, factor * cum_spread[it])], err


def check_parity_with_decisions(hip_T, ora_T, div, pose_error, tol=1e-4, tol_after_divergence=1e-3, what="", spread_T=None, ref_spread=None):
    """Per-iterate parity rule.  An iterate may leave the 1e-4 bar only if it is EXPLAINED:
      * from the iterate of the first differing accept / guard decision on (a flipped decision is a different -- equally legitimate --
        trajectory of the reference's own discontinuous algorithm), the looser bound applies;
      * before that, only up to 3x the oracle's OWN summation-order spread at that iterate (spread_T = the oracle's iterates with
        sequential fp32 sums, ora_T with exactly rounded sums): on a window where the reference's arithmetic itself is not determined
        to 1e-4 -- a dense-only two-frame window, say -- round-off alone, amplified by five PCG steps on an ill-conditioned system,
        is that large, and the reference's float atomics land in an arbitrary order.
    Round 6: where the reference's OWN spread on the window is known (ref_spread [G], cumulative: reference_licence above -- the reference's code under other legal
    atomic orders and under its own fast-math flags) it REPLACES the oracle's summation spread as the floor: the licence is what the reference does to itself.
    Returns (worst before the divergence, worst after)."""
    G, N = ora_T.shape[0], ora_T.shape[1]
    first = div[0] if div is not None else G
    wb = wa = 0.0
    for it in range(G):
        e = max(max(pose_error(hip_T[it, k], ora_T[it, k])) for k in range(N))
        if it < first:
            wb = max(wb, e)
            floor = 0.0 if spread_T is None else max(max(pose_error(spread_T[it2, k], ora_T[it2, k])) for it2 in range(it + 1) for k in range(N))
            if ref_spread is not None:
                floor = max(floor, float(ref_spread[it]))      # (both given: the larger of the two -- the oracle's own spread depends on the box's OpenMP thread count)
            assert e < max(tol, 3.0 * floor), (f"{what}: iterate {it} differs by {e:.2e} although every decision so far was identical "
                                               f"(first divergence: {div}; the {'reference' if ref_spread is not None else 'oracle'}'s own spread up to here: {floor:.2e})")
        else:
            wa = max(wa, e)
            assert e < tol_after_divergence, f"{what}: iterate {it} differs by {e:.2e} after divergence {div}"
    return wb, wa
