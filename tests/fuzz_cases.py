"""The seeded random windows of tests/tools/fuzz_parity.py, generated WITHOUT a GPU so that CPU-side tools (the reference's self-spread,
tests/tools/reference_self_spread.py) see exactly the windows the GPU-side sweep solves.  Test infrastructure."""
import numpy as np

from bundletrack_amd import synthetic as S


def fuzz_cases(n_cases, only=None):
    """Yields (case, pb, corr, meta): window sizes 2 ... 9, 0 ... 500 correspondences per pair with ragged pairs (one emptied, one thinned by
    invalidated entries: EntryJ::isValid, imgIdx_i == 0xFFFFFFFF), object-masked or fully valid frames, initial perturbations 0.5 / 2 / 4 degrees.
    Every case draws its random numbers whether it is yielded or not, so `only` selects a case without changing it."""
    rng = np.random.default_rng(20260925)
    for case in range(n_cases):
        K = int(rng.integers(2, 10))
        m = int(rng.choice([0, 5, 60, 200, 500]))
        background = bool(rng.integers(0, 2))
        perturb = float(rng.choice([0.5, 2.0, 4.0]))
        pb = S.make_problem(K, m, seed=9000 + case, background=background, perturb_deg=perturb) if (only is None or only == case) else None
        # the ragged-pair draws depend on the problem's pair counts, which depend only on (K, m)
        n_pairs = K * (K - 1) // 2
        counts = np.full(n_pairs, m, np.int64) if pb is None else np.array(pb.n_match_per_pair, np.int64).copy()
        corr = None if pb is None else pb.corr.copy()
        if m and len(counts) > 1:
            off = np.concatenate([[0], np.cumsum(counts)])
            kill = int(rng.integers(0, len(counts)))
            thin = int(rng.integers(0, len(counts)))
            sel = off[thin] + rng.choice(max(1, counts[thin]), size=max(1, counts[thin] // 3), replace=False)
            if corr is not None:
                corr["imgIdx_i"][off[kill]:off[kill + 1]] = 0xFFFFFFFF
                corr["imgIdx_i"][sel[sel < off[thin + 1]]] = 0xFFFFFFFF
        if pb is None:
            continue
        yield case, pb, corr, {"case": case, "K": K, "corr_per_pair": m, "background": background, "perturb_deg": perturb}
