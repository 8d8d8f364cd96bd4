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

    def __init__(self, hip_opt, oracle_opt, pose_error):
        self.hip, self.ora, self.pose_error = hip_opt, oracle_opt, pose_error
        self.diffs = []

    def optimizeFrames(self, global_corres, n_match_per_pair, n_frames, H, W, depths, colors, normals, poses, K):
        ref = np.array(poses, np.float32, copy=True)
        self.ora.optimizeFrames(global_corres, n_match_per_pair, n_frames, H, W, depths, colors, normals, ref, K)
        self.hip.optimizeFrames(global_corres, n_match_per_pair, n_frames, H, W, depths, colors, normals, poses, K)
        self.diffs.append(max(max(self.pose_error(poses[k], ref[k])) for k in range(n_frames)))
