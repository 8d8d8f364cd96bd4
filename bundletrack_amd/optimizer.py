"""Host-side mirror of the reference's optimiser interface on top of the C ABI.

`OptimizerGpu.optimizeFrames` keeps the name, argument order and meaning of
src/cuda/LossGPU.h:50 (std::vector<EntryJ>, n_match_per_pair, n_frames, H, W, depths_gpu,
colors_gpu, normals_gpu, poses&, K).  Device buffers are torch CUDA tensors (torch is the
device-memory plumbing here, nothing more); everything numerical happens inside libbtba.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import ENTRYJ_DTYPE, Params, Stats, TraceLayout, check, default_params, lib


def _torch():
    import torch
    return torch


class Workspace:
    """One HIP stream + grow-only device scratch (btba_workspace).  By default it runs on
    torch's current stream so torch tensors and events order naturally with the solver."""

    def __init__(self, stream: int | None = None, use_torch_stream: bool = True):
        h = C.c_void_p()
        if stream is None and use_torch_stream:
            torch = _torch()
            if not torch.cuda.is_available():
                raise RuntimeError("bundletrack_amd needs a GPU: torch.cuda.is_available() is False (no CPU fallback)")
            # torch's default stream is the NULL stream (handle 0): it must be used AS IS, otherwise the workspace
            # would run on a private non-blocking stream that does not order with torch's copies and kernels.
            check(lib().btba_workspace_create_on_stream(C.byref(h), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "btba_workspace_create_on_stream")
        else:
            check(lib().btba_workspace_create(C.byref(h), C.c_void_p(stream) if stream else None), "btba_workspace_create")
        self._h = h

    @property
    def handle(self):
        return self._h

    def sync(self):
        check(lib().btba_workspace_sync(self._h), "btba_workspace_sync")

    def set_option(self, option: int, value: int):
        """Developer / tuning switch of this workspace (_lib.OPT_*, include/btba.h BTBA_OPT_*): schedules and equivalent code paths only."""
        check(lib().btba_workspace_set_option(self._h, int(option), int(value)), "btba_workspace_set_option")

    def wait_stream(self, stream: int | None):
        """Work enqueued on the workspace after this call waits (on the device) for what `stream` (a raw HIP stream handle,
        None = the default stream) holds now: inputs produced on another stream."""
        check(lib().btba_workspace_wait_stream(self._h, C.c_void_p(stream) if stream else None), "btba_workspace_wait_stream")

    def signal_stream(self, stream: int | None):
        """`stream` waits (on the device) for what the workspace has enqueued so far: results consumed on another stream."""
        check(lib().btba_workspace_signal_stream(self._h, C.c_void_p(stream) if stream else None), "btba_workspace_signal_stream")

    def live_blocks(self) -> int:
        """8 x 8 pixel blocks walked by the dense sweeps since set_option(OPT_COUNT_LIVE, 1) (btba_workspace_live_blocks)."""
        v = C.c_uint64(0)
        check(lib().btba_workspace_live_blocks(self._h, C.byref(v)), "btba_workspace_live_blocks")
        return int(v.value)

    def collect_stats(self) -> dict:
        s = Stats()
        check(lib().btba_collect_stats(self._h, C.byref(s)), "btba_collect_stats")
        return s.as_dict()

    def close(self):
        if self._h:
            lib().btba_workspace_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _dev_ptr(t, what: str = "tensor"):
    """Raw device address of a tensor handed across the C ABI, which sees plain pointers: the tensor must be a dense
    row-major CUDA tensor of 4-byte (or byte) elements -- a strided view would be read as garbage without any error."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError(f"{what}: expected a CUDA tensor, got device {t.device}")
    if not t.is_contiguous():
        raise ValueError(f"{what}: expected a contiguous (row-major) tensor, got shape {tuple(t.shape)} with strides {tuple(t.stride())}; call .contiguous()")
    if t.element_size() not in (1, 4):
        raise ValueError(f"{what}: expected float32 / int32 / uint8 data, got {t.dtype}")
    return t.data_ptr()


def _ptr_array(tensors, what: str = "frame map"):
    arr = (C.c_void_p * len(tensors))()
    for k, t in enumerate(tensors):
        arr[k] = _dev_ptr(t, f"{what} {k}")
    return arr


class OptimizerGpu:
    """Drop-in for the reference class of the same name (src/cuda/LossGPU.h:40-52).

    The reference reads its parameters from a YAML node by string key at every use site;
    here `yml` may be a nested dict with the same keys (bundle.num_iter_outter, ...,
    p2p.max_dist, p2p.max_normal_angle) or None for the shipping defaults."""

    def __init__(self, yml: dict | None = None, workspace: Workspace | None = None, keyed_correspondences: bool = False, **overrides):
        self.params = default_params()
        self.keyed_correspondences = bool(keyed_correspondences)      # with frame_keys: keep pair segments on the device (BTBA_FLAG_KEYED_CORR)
        if yml:
            b, p = yml.get("bundle", {}), yml.get("p2p", {})
            if "num_iter_outter" in b: self.params.n_gn_iters = int(b["num_iter_outter"])
            if "num_iter_inner" in b: self.params.n_pcg_iters = int(b["num_iter_inner"])
            if "robust_delta" in b: self.params.robust_delta = float(b["robust_delta"])
            if "image_downscale" in b: self.params.image_downscale = float(b["image_downscale"])
            if "max_dist" in p: self.params.dense_dist_thresh = float(p["max_dist"])
            if "max_normal_angle" in p:
                self.params.dense_normal_thresh = float(np.cos(float(p["max_normal_angle"]) / 180.0 * np.pi))
        for k, v in overrides.items():
            setattr(self.params, k, v)
        self.workspace = workspace
        self.last_stats: dict | None = None

    def optimizeFrames(self, global_corres, n_match_per_pair, n_frames, H, W, depths_gpu, colors_gpu, normals_gpu, poses, K,
                       dense_pairs=None, frame_keys=None):
        """global_corres: ENTRYJ_DTYPE array; depths_gpu[k]: CUDA float32 [H,W]; normals_gpu[k]: CUDA float32
        [H,W,4]; colors_gpu: ignored (weight 0 in the reference, SBA.cpp:32); poses: float32 [n_frames,4,4]
        camera->model, updated IN PLACE like the reference's non-const reference argument; K: [3,3].
        frame_keys (optional, one stable integer id per frame, e.g. Frame::_id): keep the frames' caches in the
        workspace across calls (btba_optimize_frames_keyed) -- only frames not seen before are cached."""
        del colors_gpu
        corr = np.ascontiguousarray(global_corres, ENTRYJ_DTYPE)
        if len(depths_gpu) != n_frames or len(normals_gpu) != n_frames:
            raise ValueError("need one depth and one normal buffer per frame")
        for d, n in zip(depths_gpu, normals_gpu):
            if not (d.is_cuda and n.is_cuda and d.dtype == _torch().float32 and n.dtype == _torch().float32 and d.is_contiguous() and n.is_contiguous()):
                raise ValueError("depth/normal buffers must be contiguous float32 CUDA tensors")
            if d.numel() != H * W or n.numel() != 4 * H * W:
                raise ValueError("depth must hold H*W floats and normals H*W float4")
        Kf = np.ascontiguousarray(K, np.float32).reshape(9)
        nm = None if n_match_per_pair is None else np.ascontiguousarray(n_match_per_pair, np.int32).reshape(-1)
        if nm is not None and nm.shape[0] != n_frames * (n_frames - 1) // 2:
            nm = None         # the reference stores this vector and never reads it (SBA.cpp:85): any other length is legal there and is ignored here
        P = np.ascontiguousarray(poses, np.float32).reshape(n_frames, 16).copy()
        dp = None
        if dense_pairs is not None:
            dp = np.ascontiguousarray(dense_pairs, np.int32).reshape(-1, 2)
        st = Stats()
        dptr, nptr = _ptr_array(depths_gpu), _ptr_array(normals_gpu)
        head = (self.workspace.handle if self.workspace else None, C.byref(self.params), n_frames, H, W, Kf.ctypes.data,
                corr.ctypes.data if corr.shape[0] else None, corr.shape[0], nm.ctypes.data if nm is not None else None,
                C.cast(dptr, C.c_void_p), C.cast(nptr, C.c_void_p))
        tail = (dp.ctypes.data if dp is not None else None, dp.shape[0] if dp is not None else 0, P.ctypes.data, C.byref(st))
        if frame_keys is None:
            rc = lib().btba_optimize_frames(*head, *tail)
        else:
            keys = np.ascontiguousarray(frame_keys, np.uint64)
            if keys.shape != (n_frames,):
                raise ValueError("need one key per frame")
            if self.workspace is None:
                raise ValueError("the persistent frame cache lives in a Workspace: construct OptimizerGpu(workspace=...)")
            flags0 = self.params.flags
            if self.keyed_correspondences:
                self.params.flags |= _lib.FLAG_KEYED_CORR
            try:
                rc = lib().btba_optimize_frames_keyed(*head, keys.ctypes.data, *tail)
            finally:
                self.params.flags = flags0
        check(rc, "btba_optimize_frames")
        self.last_stats = st.as_dict()
        out = P.reshape(n_frames, 4, 4)
        if isinstance(poses, np.ndarray):
            poses[...] = out.reshape(poses.shape)          # the reference's in/out `poses&`
        else:                                              # a list of 4x4 arrays (std::vector<Eigen::Matrix4f> in the reference): update each element
            for k in range(n_frames):
                if isinstance(poses[k], np.ndarray):
                    poses[k][...] = out[k]
                else:
                    poses[k] = out[k].copy()
        return poses


def frame_cache_evict(ws: Workspace, key: int) -> None:
    """Forget one frame kept under `key` (a tracker that drops a frame and may reuse its id)."""
    check(lib().btba_frame_cache_evict(ws.handle, C.c_uint64(int(key))), "btba_frame_cache_evict")


def frame_cache_clear(ws: Workspace) -> None:
    """Drop every frame kept by optimizeFrames(..., frame_keys=...) in this workspace."""
    check(lib().btba_frame_cache_clear(ws.handle), "btba_frame_cache_clear")


def build_cache(ws: Workspace, depths_gpu, normals_gpu, H, W, K, image_downscale=4.0):
    """CUDACache::storeFrame for all frames in one launch.  Returns (campos, normals, n_valid, intr):
    CUDA tensors [N,Hd,Wd,4], [N,Hd,Wd,4], int32 [N] and a float32 numpy (fx,fy,cx,cy)."""
    torch = _torch()
    N = len(depths_gpu)
    Wd, Hd = int(W / image_downscale), int(H / image_downscale)
    dev = depths_gpu[0].device
    campos = torch.empty((N, Hd, Wd, 4), dtype=torch.float32, device=dev)
    normals = torch.empty((N, Hd, Wd, 4), dtype=torch.float32, device=dev)
    nvalid = torch.zeros((N,), dtype=torch.int32, device=dev)
    intr = np.zeros(4, np.float32)
    Kf = np.ascontiguousarray(K, np.float32).reshape(9)
    dptr, nptr = _ptr_array(depths_gpu), _ptr_array(normals_gpu)
    check(lib().btba_build_cache(ws.handle, N, H, W, Kf.ctypes.data, float(image_downscale), C.cast(dptr, C.c_void_p), C.cast(nptr, C.c_void_p),
                                 _dev_ptr(campos, "campos"), _dev_ptr(normals, "normals"), _dev_ptr(nvalid, "nvalid"), intr.ctypes.data), "btba_build_cache")
    return campos, normals, nvalid, intr


def build_cache_zn(ws: Workspace, depths_gpu, normals_gpu, H, W, K, image_downscale=4.0):
    """Compact frame cache: float4 (z, nx, ny, nz) per downsampled pixel.  Returns (zn [N,Hd,Wd,4], n_valid, intr)."""
    torch = _torch()
    N = len(depths_gpu)
    Wd, Hd = int(W / image_downscale), int(H / image_downscale)
    dev = depths_gpu[0].device
    zn = torch.empty((N, Hd, Wd, 4), dtype=torch.float32, device=dev)
    nvalid = torch.zeros((N,), dtype=torch.int32, device=dev)
    intr = np.zeros(4, np.float32)
    Kf = np.ascontiguousarray(K, np.float32).reshape(9)
    dptr, nptr = _ptr_array(depths_gpu), _ptr_array(normals_gpu)
    check(lib().btba_build_cache_zn(ws.handle, N, H, W, Kf.ctypes.data, float(image_downscale), C.cast(dptr, C.c_void_p), C.cast(nptr, C.c_void_p),
                                    _dev_ptr(zn, "zn"), _dev_ptr(nvalid, "nvalid"), intr.ctypes.data), "btba_build_cache_zn")
    return zn, nvalid, intr


def pack_zn(ws: Workspace, campos, normals):
    """Reference-layout float4 caches -> compact cache (camPos.xy are re-derived from z by the solver)."""
    torch = _torch()
    zn = torch.empty_like(campos)
    check(lib().btba_pack_zn(ws.handle, campos.numel() // 4, _dev_ptr(campos, "campos"), _dev_ptr(normals, "normals"), _dev_ptr(zn, "zn")), "btba_pack_zn")
    return zn


@dataclass
class TraceView:
    layout: TraceLayout
    data: np.ndarray       # [B, n_gn, record]
    n_frames: int
    n_dense_pairs: int
    n_pcg: int

    def _get(self, off, size, shape):
        return self.data[..., off:off + size].reshape(self.data.shape[:2] + shape)

    @property
    def x_after(self): return self._get(self.layout.off_x, 6 * self.n_frames, (self.n_frames, 6))
    @property
    def T_after(self): return self._get(self.layout.off_T, 16 * self.n_frames, (self.n_frames, 4, 4))
    @property
    def rhs(self): return self._get(self.layout.off_rhs, 6 * self.n_frames, (self.n_frames, 6))
    @property
    def precond(self): return self._get(self.layout.off_precond, 6 * self.n_frames, (self.n_frames, 6))
    @property
    def pcg_scalars(self): return self._get(self.layout.off_pcg, 4 * self.n_pcg, (self.n_pcg, 4))
    @property
    def delta(self): return self._get(self.layout.off_delta, 6 * self.n_frames, (self.n_frames, 6))
    @property
    def dense_pair(self): return self._get(self.layout.off_dense_pair, 28 * self.n_dense_pairs, (self.n_dense_pairs, 28))
    @property
    def clk(self): return self._get(self.layout.off_clk, 8, (8,))
    @property
    def A(self):
        n = 6 * self.n_frames
        return self._get(self.layout.off_A, n * n, (n, n))


class BatchSolver:
    """Batched, device-resident solve: many independent tracking instances in one grid
    (btba_solve_batch; the solveBundlingStub seam of the reference, SolverBundling.cu:931)."""

    def __init__(self, workspace: Workspace | None = None, **param_overrides):
        self.ws = workspace or Workspace()
        self.params = default_params(**param_overrides)

    def set_iteration_weights(self, sparse=None, dense=None):
        """Per-iteration weights of the two terms (btba_params.weights_*_per_iter; input.weightsSparse / weightsDenseDepth of the reference's
        solveBundlingStub seam): sequences of n_gn_iters floats, or None for the scalar weight in every iteration."""
        n = int(self.params.n_gn_iters)
        self._w_it = []                                     # keeps the host arrays alive as long as the params point at them
        for name, w in (("weights_sparse_per_iter", sparse), ("weights_dense_per_iter", dense)):
            if w is None:
                setattr(self.params, name, None)
                continue
            a = np.ascontiguousarray(w, np.float32)
            if a.shape != (n,):
                raise ValueError(f"{name}: expected {n} weights")
            self._w_it.append(a)
            setattr(self.params, name, a.ctypes.data)
        self.params.n_weights_per_iter = n if self._w_it else 0

    def _check_iteration_weights(self):
        """btba_params carries raw host pointers and no length: the arrays handed to set_iteration_weights must still hold n_gn_iters weights when a
        solve is enqueued (a caller that raises params.n_gn_iters afterwards would make the library read past their end)."""
        n = int(self.params.n_gn_iters)
        arrays = getattr(self, "_w_it", [])
        for name in ("weights_sparse_per_iter", "weights_dense_per_iter"):
            ptr = getattr(self.params, name)
            if not ptr:
                continue
            a = next((w for w in arrays if w.ctypes.data == ptr), None)
            if a is None or a.shape != (n,):
                raise ValueError(f"{name}: set for {'?' if a is None else a.shape[0]} iterations, params.n_gn_iters is {n} -- call set_iteration_weights again")

    @staticmethod
    def pack_correspondences(corr_list, n_frames):
        """Host: bucket each instance's EntryJ pair-major; returns (corr [B, stride] ENTRYJ, offsets [B,P+1] u32, max_per_pair)."""
        P = n_frames * (n_frames - 1) // 2
        packed = [_lib.bucket_correspondences(c, n_frames) for c in corr_list]
        stride = max(1, max(p[0].shape[0] for p in packed))
        corr = np.zeros((len(packed), stride), ENTRYJ_DTYPE)
        corr["imgIdx_i"] = 0xFFFFFFFF
        corr["imgIdx_j"] = 0xFFFFFFFF
        offs = np.zeros((len(packed), P + 1), np.uint32)
        mx = 0
        for b, (c, o) in enumerate(packed):
            corr[b, : c.shape[0]] = c
            offs[b] = o
            if P:
                mx = max(mx, int(np.diff(o.astype(np.int64)).max()))
        return corr, offs, mx

    def solve(self, campos, normals, intr, corr_dev, pair_offsets_dev, max_corr_per_pair, poses_dev, dense_pairs=None, trace=False):
        """campos/normals: CUDA float32 [B,N,Hd,Wd,4]; corr_dev: CUDA uint8 view of EntryJ [B,stride,32];
        pair_offsets_dev: CUDA int32/uint32 [B,P+1]; poses_dev: CUDA float32 [B,N,4,4] updated in place.
        Asynchronous; call .ws.sync() / .ws.collect_stats().  Returns a device trace tensor or None."""
        torch = _torch()
        self._check_iteration_weights()
        B, N, Hd, Wd = campos.shape[:4]
        intr = np.ascontiguousarray(intr, np.float32)
        stride = corr_dev.shape[1] if corr_dev is not None else 0
        dp = None
        npd = N * (N - 1) // 2
        if dense_pairs is not None:
            dp = np.ascontiguousarray(dense_pairs, np.int32).reshape(-1, 2)
            npd = dp.shape[0]
        tr = None
        L = TraceLayout()
        lib().btba_trace_layout_get(N, npd if self.params.weight_dense_depth > 0 else 0, self.params.n_pcg_iters, C.byref(L))
        if trace:
            self.params.flags |= _lib.FLAG_TRACE
            tr = torch.zeros((B, self.params.n_gn_iters, L.record_floats), dtype=torch.float32, device=campos.device)
        else:
            self.params.flags &= ~_lib.FLAG_TRACE
        rc = lib().btba_solve_batch(
            self.ws.handle, C.byref(self.params), B, N, Hd, Wd, intr.ctypes.data,
            _dev_ptr(campos, "campos"), _dev_ptr(normals, "normals"),
            _dev_ptr(corr_dev, "corr_dev"), stride,
            _dev_ptr(pair_offsets_dev, "pair_offsets_dev"), int(max_corr_per_pair),
            dp.ctypes.data if dp is not None else None, dp.shape[0] if dp is not None else 0,
            _dev_ptr(poses_dev, "poses_dev"), _dev_ptr(tr, "tr"))
        check(rc, "btba_solve_batch")
        self._last_layout = (L, N, npd if self.params.weight_dense_depth > 0 else 0)
        return tr

    def cache_aux(self, zn, valid_lists=False):
        """What the solve derives from the compact caches alone (btba_zn_aux): the per-8x8-block depth ranges (Hd, Wd multiples of 8)
        and, on request, every frame's ordered list of valid pixels (for BTBA_FLAG_COMPACTION).  Part of the frame cache: built once
        per set of frames and passed to solve_zn(aux=...).  zn CUDA float32 [..., Hd, Wd, 4]; returns a dict of CUDA tensors."""
        torch = _torch()
        Hd, Wd = int(zn.shape[-3]), int(zn.shape[-2])
        n = int(np.prod(zn.shape[:-3]))
        aux = {}
        if Hd % 8 == 0 and Wd % 8 == 0:
            aux["block_ranges"] = torch.empty((n, (Hd // 8) * (Wd // 8), 2), dtype=torch.float32, device=zn.device)
            check(lib().btba_zn_block_ranges(self.ws.handle, n, Hd, Wd, _dev_ptr(zn, "zn"), _dev_ptr(aux["block_ranges"], "block_ranges")), "btba_zn_block_ranges")
        if valid_lists:
            aux["valid_lists"] = torch.empty((n, Hd * Wd), dtype=torch.int32, device=zn.device)
            aux["valid_counts"] = torch.empty((n,), dtype=torch.int32, device=zn.device)
            check(lib().btba_zn_valid_lists(self.ws.handle, n, Hd, Wd, _dev_ptr(zn, "zn"), _dev_ptr(aux["valid_lists"], "valid_lists"), _dev_ptr(aux["valid_counts"], "valid_counts")),
                  "btba_zn_valid_lists")
        return aux

    def pack_correspondences24(self, corr_dev, pair_offsets_dev, max_corr_per_pair, n_frames, check_order=False, out=None):
        """Device-resident EntryJ [B, stride, 32] (uint8 view) -> 24-byte records, float32 [groups of 64 entries, 3 planes, 64, 2]
        (btba_pack_correspondences24; entry E = b * stride + e sits in group E // 64 at column E % 64): what a
        batch that stays on the device hands to solve_zn(aux={"corr24": ...}).  check_order: also return a device int32 flag that is 1 when an
        entry does not belong to the pair of its segment."""
        torch = _torch()
        B, stride = int(corr_dev.shape[0]), int(corr_dev.shape[1])
        if out is None:                # (out: a tensor from an earlier call on an array of the same shape -- a caller that re-packs fresh matches every solve reuses it)
            out = torch.zeros((-(-(B * stride) // 64), 3, 64, 2), dtype=torch.float32, device=corr_dev.device)      # groups of 64 entries x 3 planes of float2
        flag = torch.zeros((1,), dtype=torch.int32, device=corr_dev.device) if check_order else None
        check(lib().btba_pack_correspondences24(self.ws.handle, B, int(n_frames), _dev_ptr(corr_dev, "corr_dev"), stride, _dev_ptr(pair_offsets_dev, "pair_offsets_dev"),
                                                int(max_corr_per_pair), _dev_ptr(out, "corr24"), _dev_ptr(flag, "order_flag")), "btba_pack_correspondences24")
        return (out, flag) if check_order else out

    def solve_zn(self, zn, H, W, K, corr_dev, pair_offsets_dev, max_corr_per_pair, poses_dev, dense_pairs=None, trace=False, aux=None, corr_stride=None):
        """Same as solve() on compact caches: zn CUDA float32 [B,N,Hd,Wd,4] = (z, nx, ny, nz); H, W, K = the FULL-resolution
        frame geometry the caches were built from (Hd = H / image_downscale).  aux: the result of cache_aux(zn) for callers that
        keep their caches across solves (None: derived inside every solve).  With aux["corr24"] (pack_correspondences24) corr_dev may be None;
        corr_stride then gives the entries per instance block of the array that was packed."""
        torch = _torch()
        self._check_iteration_weights()
        B, N = zn.shape[:2]
        Kf = np.ascontiguousarray(K, np.float32).reshape(9)
        stride = corr_dev.shape[1] if corr_dev is not None else int(corr_stride or 0)
        dp = None
        npd = N * (N - 1) // 2
        if dense_pairs is not None:
            dp = np.ascontiguousarray(dense_pairs, np.int32).reshape(-1, 2)
            npd = dp.shape[0]
        tr = None
        L = TraceLayout()
        lib().btba_trace_layout_get(N, npd if self.params.weight_dense_depth > 0 else 0, self.params.n_pcg_iters, C.byref(L))
        if trace:
            self.params.flags |= _lib.FLAG_TRACE
            tr = torch.zeros((B, self.params.n_gn_iters, L.record_floats), dtype=torch.float32, device=zn.device)
        else:
            self.params.flags &= ~_lib.FLAG_TRACE
        za = None
        if aux:
            za = _lib.ZnAux(_dev_ptr(aux.get("block_ranges"), "block_ranges"), _dev_ptr(aux.get("valid_lists"), "valid_lists"), _dev_ptr(aux.get("valid_counts"), "valid_counts"),
                            _dev_ptr(aux.get("corr24"), "corr24"))
        rc = lib().btba_solve_batch_zn_aux(
            self.ws.handle, C.byref(self.params), B, N, int(H), int(W), Kf.ctypes.data, _dev_ptr(zn, "zn"), C.byref(za) if za is not None else None,
            _dev_ptr(corr_dev, "corr_dev"), stride,
            _dev_ptr(pair_offsets_dev, "pair_offsets_dev"), int(max_corr_per_pair),
            dp.ctypes.data if dp is not None else None, dp.shape[0] if dp is not None else 0,
            _dev_ptr(poses_dev, "poses_dev"), _dev_ptr(tr, "tr"))
        check(rc, "btba_solve_batch_zn_aux")
        self._last_layout = (L, N, npd if self.params.weight_dense_depth > 0 else 0)
        return tr

    def trace_view(self, tr) -> TraceView:
        L, N, npd = self._last_layout
        self.ws.sync()
        return TraceView(L, tr.cpu().numpy(), N, npd, self.params.n_pcg_iters)


# ---- the step before the boundary: Frame::processDepth / Frame::depthToCloudAndNormals ----
DEPTH_PROCESSING_DEFAULTS = dict(erode_radius=1, erode_diff=0.001, erode_ratio=0.8, bf_radius=2, sigma_d=2.0, sigma_r=100000.0)   # config_ycbineoat.yml:9-16


def process_depth(ws: Workspace, depth_gpu, **kw):
    """Frame::processDepth (src/Frame.cpp:152-180) on a CUDA float32 [H,W] depth map; returns a new tensor."""
    torch = _torch()
    p = dict(DEPTH_PROCESSING_DEFAULTS); p.update(kw)
    H, W = depth_gpu.shape
    out = torch.empty_like(depth_gpu)
    check(lib().btba_process_depth(ws.handle, H, W, _dev_ptr(depth_gpu, "depth_gpu"), _dev_ptr(out, "out"), int(p["erode_radius"]), float(p["erode_diff"]), float(p["erode_ratio"]),
                                   int(p["bf_radius"]), float(p["sigma_d"]), float(p["sigma_r"])), "btba_process_depth")
    return out


def depth_to_normals(ws: Workspace, depth_gpu, K, want_xyz=False):
    """Frame::depthToCloudAndNormals (src/Frame.cpp:182-233): returns normals float4 [H,W,4] (and the xyz map)."""
    torch = _torch()
    H, W = depth_gpu.shape
    normals = torch.empty((H, W, 4), dtype=torch.float32, device=depth_gpu.device)
    xyz = torch.empty((H, W, 4), dtype=torch.float32, device=depth_gpu.device) if want_xyz else None
    Kf = np.ascontiguousarray(K, np.float32).reshape(9)
    check(lib().btba_depth_to_normals(ws.handle, H, W, Kf.ctypes.data, _dev_ptr(depth_gpu, "depth_gpu"), _dev_ptr(normals, "normals"), _dev_ptr(xyz, "xyz")), "btba_depth_to_normals")
    return (normals, xyz) if want_xyz else normals
