"""bundletrack_amd -- MI355X-native pose-graph bundle adjustment behind BundleTrack's
OptimizerGpu::optimizeFrames boundary.  The compute path is libbtba.so
(hand-written HIP for gfx950); importing this package never falls back to CPU code."""
from . import _lib  # noqa: F401
from ._lib import ENTRYJ_DTYPE, BtbaError, default_params  # noqa: F401

__all__ = ["ENTRYJ_DTYPE", "BtbaError", "default_params"]
