"""ctypes binding of libbtba.so (the C ABI of include/btba.h).

The HIP library is the product; this module only loads it.  It fails loudly when the
library is missing -- there is no CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, "libbtba.so")
SRC_DIR = os.path.join(_PKG, "csrc")
HEADER = os.path.join(_ROOT, "include", "btba.h")

BTBA_OK, BTBA_EINVAL, BTBA_EHIP, BTBA_ENUMERIC, BTBA_ENOMEM, BTBA_ESCHED = 0, 1, 2, 3, 4, 5
PAIRS_TARGET_LOWER, PAIRS_TARGET_MORE_VALID, PAIRS_EXPLICIT, PAIRS_TARGET_HIGHER = 0, 1, 2, 3
REDUCE_DETERMINISTIC, REDUCE_ATOMIC = 0, 1
RANSAC_REFERENCE_SVD, RANSAC_HORN = 0, 1
RANSAC_DRAW_HASH = 0x100      # ORed into `hypothesis`: counter-hash sample triples instead of the reference's cuRAND XORWOW stream
FLAG_TRACE, FLAG_TIME_KERNELS = 1, 2
FLAG_OVERLAP, FLAG_NO_FUSE, FLAG_KEYED_CORR, FLAG_FLOAT4_CACHE, FLAG_NO_COMPACTION, FLAG_COMPACTION, FLAG_TIME_SAMPLED = 32, 64, 4096, 256, 512, 1024, 2048
OPT_DENSE_ORDER, OPT_TILE_MAJOR, OPT_BLOCK_WALK, OPT_BLOCK_SKIP, OPT_BIG_ASSEMBLY, OPT_OVERLAP_GROUPS, OPT_OVERLAP_EQUAL_PRIO, OPT_KEYED_CORR_MIN_BYTES, OPT_SPARSE_TAIL = 1, 2, 3, 4, 5, 6, 7, 8, 9
OPT_CHAIN, OPT_CHAIN_SPARSE_PERIOD, OPT_CHAIN_TIMEOUT_MS, OPT_COUNT_LIVE, OPT_RELAYOUT, OPT_CORR_NONTEMPORAL, OPT_SOLVE_SMALL = 10, 11, 12, 13, 14, 15, 16

ENTRYJ_DTYPE = np.dtype(
    [("imgIdx_i", "<u4"), ("imgIdx_j", "<u4"), ("pos_i", "<f4", (3,)), ("pos_j", "<f4", (3,))]
)

EXPORTED_SYMBOLS = [
    "btba_params_default", "btba_strerror", "btba_last_hip_error", "btba_version",
    "btba_workspace_create", "btba_workspace_create_on_stream", "btba_workspace_destroy", "btba_workspace_sync",
    "btba_workspace_wait_stream", "btba_workspace_signal_stream", "btba_workspace_set_option", "btba_workspace_live_blocks",
    "btba_optimize_frames", "btba_optimize_frames_keyed", "btba_frame_cache_clear", "btba_frame_cache_evict", "btba_ransac_pairs", "btba_ransac_pairs_ex", "btba_ransac_reference_uniforms", "btba_build_cache", "btba_solve_batch", "btba_solve_cached", "btba_collect_stats",
    "btba_trace_layout_get", "btba_bucket_correspondences",
    "btba_matrices_to_poses", "btba_poses_to_matrices",
    "btba_process_depth", "btba_depth_to_normals",
    "btba_build_cache_zn", "btba_pack_zn", "btba_solve_batch_zn", "btba_zn_block_ranges", "btba_zn_valid_lists", "btba_solve_batch_zn_aux", "btba_pack_correspondences24",
]


class Params(C.Structure):
    _fields_ = [
        ("n_gn_iters", C.c_int32), ("n_pcg_iters", C.c_int32),
        ("robust_delta", C.c_float), ("dense_dist_thresh", C.c_float), ("dense_normal_thresh", C.c_float),
        ("depth_min", C.c_float), ("depth_max", C.c_float),
        ("weight_sparse", C.c_float), ("weight_dense_depth", C.c_float), ("image_downscale", C.c_float),
        ("pair_policy", C.c_int32), ("dense_tiles", C.c_int32), ("sparse_chunks", C.c_int32), ("flags", C.c_int32), ("reduction_mode", C.c_int32),
        ("weights_sparse_per_iter", C.c_void_p), ("weights_dense_per_iter", C.c_void_p),      # host float[n_gn_iters] or NULL (the scalars)
        ("n_weights_per_iter", C.c_int32),                                                     # their length (must equal n_gn_iters when either is set)
    ]


class Stats(C.Structure):
    _fields_ = [
        ("n_instances", C.c_int32), ("n_frames", C.c_int32), ("n_pairs", C.c_int32), ("n_dense_pairs", C.c_int32),
        ("n_corr", C.c_int64),
        ("dense_tiles", C.c_int32), ("sparse_chunks", C.c_int32),
        ("ms_total", C.c_float), ("ms_upload", C.c_float), ("ms_cache", C.c_float), ("ms_solve", C.c_float),
        ("ms_dense_sweep", C.c_float), ("ms_sparse_sweep", C.c_float), ("ms_system_solve", C.c_float),
        ("n_dense_launches", C.c_int32), ("n_sparse_launches", C.c_int32), ("n_solve_launches", C.c_int32),
        ("bytes_dense_alg", C.c_int64), ("bytes_sparse_alg", C.c_int64),
        ("fused_sweeps", C.c_int32), ("cache_frames_built", C.c_int32), ("corr_pairs_uploaded", C.c_int32),
        ("chain_iterations", C.c_int32),
    ]

    def as_dict(self):
        return {f[0]: getattr(self, f[0]) for f in self._fields_}


class ZnAux(C.Structure):
    """btba_zn_aux (include/btba.h): device pointers to data derived from compact caches alone."""
    _fields_ = [("block_ranges", C.c_void_p), ("valid_lists", C.c_void_p), ("valid_counts", C.c_void_p), ("corr24", C.c_void_p)]


class TraceLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("record_floats", "off_x", "off_T", "off_rhs", "off_precond", "off_pcg", "off_delta", "off_dense_pair", "off_A", "off_clk")]


class BtbaError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        msg = lib().btba_strerror(status).decode() if _lib is not None else str(status)
        extra = f" (hipError {lib().btba_last_hip_error()})" if status == BTBA_EHIP else ""
        super().__init__(f"{where}: {msg}{extra}")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR)] + [HEADER]
    newest = max(os.path.getmtime(s) for s in srcs)
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest:
        return LIB_PATH
    # -fno-slp-vectorize: the SLP pass packs neighbouring fp32 operations into v_pk_{mul,add,fma}_f32, which issue at half
    # rate on gfx950 and need register-pair shuffling (v_mov) around them; measured -14 % on the dense sweep, -7 % on
    # the sparse sweep without it.
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-pass-failed", "-fPIC", "-shared", "-fvisibility=hidden",
           "-o", LIB_PATH, os.path.join(SRC_DIR, "btba_api.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


HOST_DRIVER = os.path.join(os.path.dirname(_PKG), "tests", "cpp", "host_driver")


def build_host_cpp(force: bool = False, verbose: bool = False) -> str:
    """Compile the C++ host layer (bundletrack_amd/cpp) and its test driver against libbtba.so."""
    root = os.path.dirname(_PKG)
    srcs = [os.path.join(root, "tests", "cpp", "host_driver.cpp"), os.path.join(_PKG, "cpp", "btba_host.cpp")]
    deps = srcs + [os.path.join(_PKG, "cpp", "btba_host.hpp"), HEADER, LIB_PATH]
    if not force and os.path.exists(HOST_DRIVER) and os.path.getmtime(HOST_DRIVER) >= max(os.path.getmtime(d) for d in deps):
        return HOST_DRIVER
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-o", HOST_DRIVER] + srcs + [
        "-L" + _PKG, "-lbtba", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
        "-Wl,-rpath," + _PKG, "-Wl,-rpath,$ORIGIN/../../bundletrack_amd", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_DRIVER


_lib = None


def lib() -> C.CDLL:
    """Load libbtba.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        path = os.environ.get("BTBA_LIB_PATH", LIB_PATH)      # developer A/B of kernel builds; still no fallback
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: the HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
        # torch first: libbtba.so is linked against /opt/rocm's libamdhip64, torch brings its own copy under the same soname.  Whichever is
        # loaded first becomes THE HIP runtime of the process; with libbtba.so first, torch's kernels and ours meet a runtime torch was not
        # built for and the first launch fails with hipErrorNoDevice (seen with build() followed by smoke() in one process).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        # the structs below mirror ONE version of include/btba.h: a library built from another one must not be called (btba.h: BTBA_VERSION)
        import re
        header_version = int(re.search(r"#define BTBA_VERSION (\d+)", open(HEADER).read()).group(1))
        if "BTBA_LIB_PATH" not in os.environ and L.btba_version() != header_version:
            raise ImportError(f"{path} is version {L.btba_version()}, include/btba.h is {header_version}: rebuild (python -c 'import __graft_entry__ as g; g.build()')")
        L.btba_strerror.restype = C.c_char_p
        L.btba_strerror.argtypes = [C.c_int]
        for name in EXPORTED_SYMBOLS:
            if "BTBA_LIB_PATH" in os.environ and name in ("btba_workspace_set_option", "btba_pack_correspondences24", "btba_workspace_live_blocks") and not hasattr(L, name):
                continue               # developer A/B against a build from before version 103
            getattr(L, name)           # AttributeError if the ABI and the header drift apart
        L.btba_workspace_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p]
        L.btba_workspace_create_on_stream.argtypes = [C.POINTER(C.c_void_p), C.c_void_p]
        L.btba_workspace_destroy.argtypes = [C.c_void_p]
        L.btba_workspace_destroy.restype = None
        L.btba_workspace_sync.argtypes = [C.c_void_p]
        if hasattr(L, "btba_workspace_set_option"):
            L.btba_workspace_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        if hasattr(L, "btba_workspace_live_blocks"):
            L.btba_workspace_live_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.btba_workspace_wait_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.btba_workspace_signal_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.btba_collect_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.btba_optimize_frames.argtypes = [
            C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_void_p,
            C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Stats)]
        L.btba_optimize_frames_keyed.argtypes = [
            C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_void_p,
            C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Stats)]
        L.btba_frame_cache_clear.argtypes = [C.c_void_p]
        L.btba_frame_cache_evict.argtypes = [C.c_void_p, C.c_uint64]
        L.btba_solve_cached.argtypes = [
            C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.btba_build_cache.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.btba_solve_batch.argtypes = [
            C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.btba_trace_layout_get.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(TraceLayout)]
        L.btba_trace_layout_get.restype = None
        L.btba_bucket_correspondences.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.btba_matrices_to_poses.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.btba_poses_to_matrices.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.btba_build_cache_zn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.btba_pack_zn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.btba_solve_batch_zn.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.btba_zn_block_ranges.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        if hasattr(L, "btba_pack_correspondences24"):
            L.btba_pack_correspondences24.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.btba_zn_valid_lists.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.btba_solve_batch_zn_aux.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(ZnAux),
                                              C.c_void_p, C.c_int64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.btba_process_depth.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
        L.btba_depth_to_normals.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def check(status: int, where: str) -> None:
    if status != BTBA_OK:
        raise BtbaError(status, where)


def default_params(**kw) -> Params:
    p = Params()
    lib().btba_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def declared_symbols() -> list[str]:
    """Function names declared in include/btba.h (used by the ABI test)."""
    import re
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"^BTBA_API[^;]*?\b(btba_[a-z_0-9]+)\s*\(", txt, flags=re.M)))


def bucket_correspondences(corr: np.ndarray, n_frames: int):
    """Host helper: stable bucketing of EntryJ by canonical frame pair -> (sorted, offsets[P+1])."""
    corr = np.ascontiguousarray(corr, ENTRYJ_DTYPE)
    P = n_frames * (n_frames - 1) // 2
    out = np.zeros(max(corr.shape[0], 1), ENTRYJ_DTYPE)
    off = np.zeros(P + 1, np.uint32)
    check(lib().btba_bucket_correspondences(corr.ctypes.data, corr.shape[0], n_frames, out.ctypes.data, off.ctypes.data),
          "btba_bucket_correspondences")
    return out[: int(off[P])], off
