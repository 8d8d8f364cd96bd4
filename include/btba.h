/*
 * btba.h -- C ABI of the MI355X-native pose-graph bundle adjustment (libbtba.so).
 *
 * Drop-in boundary for BundleTrack's  OptimizerGpu::optimizeFrames
 *   (/root/reference/src/cuda/LossGPU.h:40-52, LossGPU.cu:53-139; call site src/Bundler.cpp:350-351)
 * and for the C-linkage seams underneath it
 *   (solveBundlingStub, buildVariablesToCorrespondencesTableCUDA, convertLiePosesToMatricesCU:
 *    src/cuda/Solver/CUDASolverBundling.cpp:8-16; convertMatricesToPosesCU / convertPosesToMatricesCU:
 *    src/cuda/SBA.cpp:10-13).
 *
 * Plain C: pointers, sizes, PODs.  No torch / Eigen / YAML types.  Device pointers are raw HIP
 * device addresses; `stream` arguments are hipStream_t passed as void*.  No entry point ever
 * exits, aborts or spins (the reference does all three: cutil_inline_runtime.h:261-269,
 * SolverBundling.cu:621-625): errors come back as an int status.
 */
#ifndef BTBA_H_
#define BTBA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BTBA_VERSION 105     /* 105: BTBA_OPT_SOLVE_SMALL, btba_params.n_weights_per_iter; 104: chained launch (BTBA_OPT_CHAIN*, BTBA_ESCHED, btba_stats.chain_iterations), btba_params.weights_*_per_iter, btba_workspace_live_blocks; 103: btba_params.reduction_mode, BTBA_FLAG_KEYED_CORR = 4096,
                                btba_workspace_set_option, btba_zn_aux.corr24 + btba_pack_correspondences24.  A caller built against another version's structs must not
                                call in: check btba_version() == BTBA_VERSION once after loading the library (the Python and C++ host layers do). */

#if defined(__GNUC__)
#define BTBA_API __attribute__((visibility("default")))
#else
#define BTBA_API
#endif

/* ---- status codes ---------------------------------------------------------------------- */
enum {
    BTBA_OK       = 0,
    BTBA_EINVAL   = 1,   /* bad argument (n_frames < 2, null pointer, unsorted batch correspondences, ...) */
    BTBA_EHIP     = 2,   /* a HIP runtime call failed; btba_last_hip_error() has the hipError_t */
    BTBA_ENUMERIC = 3,   /* a non-finite value reached the output poses */
    BTBA_ENOMEM   = 4,
    BTBA_ESCHED   = 5    /* a wait inside the chained launch ran into its watchdog (the device did not start the launch's workgroups in grid
                            order): the poses of that solve are invalid.  Reported by the next call that synchronises with the host
                            (btba_workspace_sync, btba_collect_stats); the workspace solves with the plain schedule from then on. */
};

/* ---- wire formats ---------------------------------------------------------------------- */
/* struct EntryJ, src/cuda/SIFTImageManager.h:44-59: 32 bytes; invalid <=> imgIdx_i == 0xFFFFFFFF.
 * pos_i / pos_j are the matched 3-D points in the camera frames of frame i / frame j
 * (Bundler.cpp:311-316: i < j, pos_i = _ptB_cam, pos_j = _ptA_cam). */
typedef struct btba_entryj {
    uint32_t imgIdx_i;
    uint32_t imgIdx_j;
    float pos_i[3];
    float pos_j[3];
} btba_entryj;

/* Dense pair orientation (the reference derives it from device
 * allocation addresses, SolverBundling.cu:25-33, so it must be an explicit choice here). */
enum {
    BTBA_PAIRS_TARGET_LOWER      = 0,  /* every i<j once, target = i (BundleFusion behaviour; default) */
    BTBA_PAIRS_TARGET_MORE_VALID = 1,  /* target = frame with more valid depth pixels, ties i<j; literal
                                          FlipJtJ semantics: the cross block vanishes when target > source */
    BTBA_PAIRS_EXPLICIT          = 2,  /* caller passes the ordered (target, source) list */
    BTBA_PAIRS_TARGET_HIGHER     = 3   /* every i<j once, target = j: what FindImageImageCorr_Kernel emits when the per-frame
                                          d_num_valid_points allocations have ASCENDING addresses in frame order (one cudaMalloc per
                                          frame in a fresh CUDACache, CUDACacheUtil.h:14) -- every dense cross block is then above
                                          the diagonal and erased by FlipJtJ_Kernel (SolverBundling.cu:49-59) */
};

/* How the sweeps' sums are reduced.  DETERMINISTIC (default): every workgroup stores its partial record, k_system_solve adds them in a
 * fixed order -- run-to-run reproducible bits.  ATOMIC: what the reference does (SolverBundlingDenseUtil.h:217-285, SolverBundling.cu
 * warpReduce + atomicAdd throughout): workgroups add into one record per frame pair with hardware float atomics, in whatever order they
 * finish -- results move in the last bits from run to run, exactly as the reference's do; no partial arrays, one reduction pass less.
 * Not available together with BTBA_FLAG_TRACE (the decision traces are defined on the reproducible sums). */
enum {
    BTBA_REDUCE_DETERMINISTIC = 0,
    BTBA_REDUCE_ATOMIC        = 1
};

enum {
    BTBA_FLAG_TRACE         = 1,    /* record per-GN-iterate trace (btba_trace_layout)                            */
    BTBA_FLAG_TIME_KERNELS  = 2,    /* bracket every sweep / solve launch with hipEvents (btba_stats)             */
    BTBA_FLAG_TIME_SAMPLED  = 2048, /* with TIME_KERNELS: bracket the launches of ONE Gauss-Newton iteration per solve only (the iteration
                                       rotates from solve to solve): 28 event records per 7-iteration solve cost ~4 % of a c3 x 32 step,
                                       4 do not; the per-launch averages in btba_stats are over the sampled launches              */
    BTBA_FLAG_OVERLAP       = 32,   /* split a batch over two streams (main + low-priority) so one half's k_system_solve
                                       overlaps the other half's sweeps; per-kernel timings then overlap too (+4 % at c3 x 32) */
    BTBA_FLAG_NO_FUSE       = 64,   /* launch the sparse and the dense sweep separately (default: ONE interleaved launch) */
    BTBA_FLAG_FLOAT4_CACHE  = 256,  /* btba_optimize_frames: build the reference-layout float4 cache instead of the compact one */
    /* 128: reserved.  It was BTBA_FLAG_FUSE ("accepted for compatibility") up to version 100 and must not acquire a meaning: ignored. */
    BTBA_FLAG_KEYED_CORR    = 4096, /* btba_optimize_frames_keyed: keep every frame PAIR's correspondence segment on the device under the
                                       pair's two frame keys and upload only segments not seen before (in a sliding window: the new
                                       frame's n_frames - 1 pairs).  CONTRACT: the correspondences of a pair do not change while both
                                       frames stay cached -- the reference never recomputes a pair's matches either (findCorres returns
                                       early when _matches holds the pair, FeatureManager.cpp:176) -- and n_match_per_pair is given.  A
                                       segment whose length changed is uploaded again (its superseded copy stays in the pool until one of the
                                       pair's frames is evicted or the cache is cleared); a cached segment is not checked for pair order
                                       again; btba_frame_cache_evict / _clear drop segments */
    BTBA_FLAG_NO_COMPACTION = 512,  /* compact cache: always walk all Wd x Hd source pixels                        */
    BTBA_FLAG_COMPACTION    = 1024  /* compact cache: walk each source frame's ordered list of pixels that carry a depth
                                       (masked scenes).  btba_optimize_frames decides by itself from the valid-pixel counts
                                       unless one of the two is set; btba_solve_batch_zn (asynchronous, no read-back) uses
                                       lists only when this flag is set */
};

/* Solver parameters.  Defaults = shipping config of the reference:
 *   config_ycbineoat.yml:23-31,63-65; SBA.cpp:27-32; CUDASolverBundling.cpp:93-98. */
typedef struct btba_params {
    int32_t n_gn_iters;           /* bundle.num_iter_outter      7      */
    int32_t n_pcg_iters;          /* bundle.num_iter_inner       5      */
    float   robust_delta;         /* bundle.robust_delta         0.005  */
    float   dense_dist_thresh;    /* p2p.max_dist                0.02   */
    float   dense_normal_thresh;  /* cos(p2p.max_normal_angle)   cos 45 deg */
    float   depth_min;            /* denseDepthMin               0.1    */
    float   depth_max;            /* denseDepthMax               9999   */
    float   weight_sparse;        /* m_localWeightsSparse        1      */
    float   weight_dense_depth;   /* m_localWeightsDenseDepth    1  (0 disables the dense term)       */
    float   image_downscale;      /* bundle.image_downscale      4      */
    int32_t pair_policy;          /* BTBA_PAIRS_*                                                       */
    int32_t dense_tiles;          /* workgroups per dense frame pair (0 = auto)                        */
    int32_t sparse_chunks;        /* workgroups per correspondence segment (0 = auto)                  */
    int32_t flags;                /* BTBA_FLAG_*                                                        */
    int32_t reduction_mode;       /* BTBA_REDUCE_*                                                      */
    /* Optional per-iteration weights, the form the solveBundlingStub seam takes them in (input.weightsSparse[nIter], input.weightsDenseDepth[nIter],
     * SolverBundling.cu:948-951; SBA.cpp:27-32 fills both with constants): HOST arrays of n_gn_iters floats >= 0, or NULL = weight_sparse /
     * weight_dense_depth in every iteration.  As in the reference, an iteration with dense weight 0 builds no dense system (useDense = false,
     * :951-953), and an iteration with sparse weight 0 still derives its Jacobi preconditioner from the correspondences (the preconditioner carries
     * no weight, SolverBundlingEquationsLie.h:107-108) while right-hand side and operator take the factor 0. */
    const float *weights_sparse_per_iter;
    const float *weights_dense_per_iter;
    int32_t n_weights_per_iter;   /* length of the arrays above as the caller allocated them: with either pointer set it must equal n_gn_iters (BTBA_EINVAL
                                   * otherwise -- the library never reads past what was stated); ignored when both are NULL.  btba_params_default: 0 */
} btba_params;

/* Timing / diagnostics filled by the solve entry points (all times in milliseconds, measured
 * with hipEvents on the workspace stream; per-kernel fields need BTBA_FLAG_TIME_KERNELS). */
/* Largest window: 85 frames, the reference's MAX_NUM_IMAGES (GlobalDefines.h:8; its solver spins forever above it,
 * SolverBundling.cu:621-625; it ships max_BA_frames = 15, config_ycbineoat.yml:27).  Up to BTBA_MAX_FRAMES_LDS frames the
 * 6N x 6N normal matrix of an instance lives in one CU's 160 KB of LDS (186 x 188 floats at N = 31) and a single wave runs
 * the PCG; larger windows keep the matrix in an L2-resident device scratch and run a 16-wave PCG (slower per iterate, same
 * arithmetic).  A window of more than BTBA_MAX_FRAMES frames is BTBA_EINVAL. */
#define BTBA_MAX_FRAMES 85
#define BTBA_MAX_FRAMES_LDS 31

typedef struct btba_stats {
    int32_t n_instances, n_frames, n_pairs, n_dense_pairs;
    int64_t n_corr;               /* total correspondences over all instances                          */
    int32_t dense_tiles, sparse_chunks;
    float ms_total;               /* whole call, host wall clock                                       */
    float ms_upload;              /* H2D of EntryJ + poses (optimize_frames only)                      */
    float ms_cache;               /* frame cache build (A3)                                            */
    float ms_solve;               /* pose-in -> pose-out region on the stream                          */
    float ms_dense_sweep;         /* sum over GN iterations of the dense Jacobian sweep kernel         */
    float ms_sparse_sweep;        /* ... of the sparse sweep kernel                                    */
    float ms_system_solve;        /* ... of the assemble + PCG + update kernel                         */
    int32_t n_dense_launches, n_sparse_launches, n_solve_launches;
    int64_t bytes_dense_alg;      /* algorithmic bytes of ONE dense sweep launch  (64 * Pd * npix * B) */
    int64_t bytes_sparse_alg;     /* algorithmic bytes of ONE sparse sweep launch (32 * C)             */
    int32_t fused_sweeps;         /* 1: sparse + dense sweeps ran as ONE launch (timed as ms_dense_sweep) */
    int32_t cache_frames_built;   /* optimize_frames: frames cached in this call (n_frames unless keyed and already cached) */
    int32_t corr_pairs_uploaded;  /* optimize_frames: frame-pair segments that crossed PCIe in this call (all P unless BTBA_FLAG_KEYED_CORR) */
    int32_t chain_iterations;     /* > 0: the solve ran as ONE chained launch carrying this many Gauss-Newton iterations (sweeps AND system solves);
                                     it is timed as ms_dense_sweep / n_dense_launches, ms_system_solve stays 0 */
} btba_stats;

/* Per-instance, per-GN-iteration trace record (floats), written when BTBA_FLAG_TRACE is set.
 * Record r = instance * n_gn_iters + iteration; record size = btba_trace_layout.record_floats.
 *   x_after     [N][6]   (rot, trans) after the update
 *   T_after     [N][16]  Exp(x_after), row-major
 *   rhs         [N][6]   (rRot, rTrans): PCG right-hand side  (frame 0 = 0)
 *   precond     [N][6]   (precRot, precTrans)                 (frame 0 = 0)
 *   pcg         [n_pcg][4]  pAp, alpha, rz_new, beta
 *   delta       [N][6]   PCG solution (deltaRot, deltaTrans)
 *   dense_pair  [Pd][28] S (21, upper triangle row-major of the 6x6 in [trans,rot] order), g (6), count
 *   A           [6N][6N] assembled normal matrix (sparse + dense), reference dense layout
 *   clk         [8]      shader-clock stamps of k_system_solve's phases (reduce, congruence, assemble, PCG, update)
 */
typedef struct btba_trace_layout {
    int64_t record_floats;
    int64_t off_x, off_T, off_rhs, off_precond, off_pcg, off_delta, off_dense_pair, off_A, off_clk;
} btba_trace_layout;

typedef struct btba_workspace btba_workspace;

/* ---- API -------------------------------------------------------------------------------- */
BTBA_API void btba_params_default(btba_params *p);
BTBA_API const char *btba_strerror(int status);
BTBA_API int btba_last_hip_error(void);
BTBA_API int btba_version(void);

/* One workspace = one HIP stream + reusable device scratch.  `stream` may be NULL (the
 * workspace then creates and owns a non-blocking stream: the caller orders it against its own streams with
 * btba_workspace_wait_stream / _signal_stream).  Re-entrant across workspaces.  A workspace lives on the device that is
 * current when it is created; every call that takes it switches to that device for its duration and restores the caller's
 * (one thread may drive several GPUs, one workspace each; buffers handed to a call must live on the workspace's device). */
BTBA_API int btba_workspace_create(btba_workspace **out, void *stream);
/* Same, but `stream` is used as given even when it is the NULL (legacy default) stream -- what a framework whose
 * "current stream" is the default stream (PyTorch) needs so that its own copies and kernels order with the solver. */
BTBA_API int btba_workspace_create_on_stream(btba_workspace **out, void *stream);
BTBA_API void btba_workspace_destroy(btba_workspace *ws);
BTBA_API int btba_workspace_sync(btba_workspace *ws);

/* Developer / tuning switches of a workspace.  None of them changes WHAT is computed (only schedules and which of two equivalent
 * code paths runs); they exist for A/B measurements and for tests that hold one path against the other.  Each also has an environment
 * variable that sets the initial value -- read once, inside btba_workspace_create*, never on the solve path. */
enum {
    BTBA_OPT_DENSE_ORDER          = 1,  /* 1 (default): dense pairs worked off heaviest first (|i - j| ascending); 0: list order.  env BTBA_NO_DENSE_ORDER */
    BTBA_OPT_TILE_MAJOR           = 2,  /* 1 (default): (band, pair) work order inside an instance; 0: (pair, band).            env BTBA_PAIR_MAJOR     */
    BTBA_OPT_BLOCK_WALK           = 3,  /* 1 (default): pinhole sweep walks 8 x 8 pixel blocks; 0: 64 x 1 strips.               env BTBA_NO_BLOCK_WALK  */
    BTBA_OPT_BLOCK_SKIP           = 4,  /* 1 (default): provably dead blocks are not walked; 0: every block is.                 env BTBA_NO_BLOCK_SKIP  */
    BTBA_OPT_BIG_ASSEMBLY         = 5,  /* 1 (default): many-workgroup reduction / assembly from 24 frames on; 0: one workgroup. env BTBA_NO_BIG_ASSEMBLY */
    BTBA_OPT_OVERLAP_GROUPS       = 6,  /* instance groups of BTBA_FLAG_OVERLAP, 1 .. 8 (default 2).                            env BTBA_GROUPS         */
    BTBA_OPT_OVERLAP_EQUAL_PRIO   = 7,  /* 1: the groups' streams get equal priority (default 0: lowest for groups >= 1).        env BTBA_GROUP_PRIO=e   */
    BTBA_OPT_SPARSE_TAIL          = 9,  /* 0 .. 256: share (x / 256) of the sparse items that close the fused sweep instead of being interleaved (fills the launch's drain); -1 (default): 256 on full frames, 0 on object-masked ones. env BTBA_SPARSE_TAIL */
    BTBA_OPT_KEYED_CORR_MIN_BYTES = 8,  /* BTBA_FLAG_KEYED_CORR is ignored below this many bytes of correspondences (default 1 MiB). env of the same name */
    BTBA_OPT_CHAIN                = 10, /* the chained launch: ALL Gauss-Newton iterations of a batch in one launch, every instance's system solve handed
                                           over inside the launch while the other instances' sweeps run (btba_kernels.hpp: k_chain).  0 (default) / -1: the
                                           plain schedule (two launches per iteration) -- it measured faster once one tile per pair became the better
                                           choice; 1: every batch the launch supports (pinhole compact cache, sparse + dense terms,
                                           <= 15 frames, no trace, deterministic sums).  Same bits as the plain schedule with the same tile count AND BTBA_OPT_SOLVE_SMALL = 0 (the in-launch
                                           solve items reproduce k_system_solve's sums); against the default plain schedule (k_solve_small: the same sums in another
                                           order) the poses agree to the 1e-4 bar, not bit for bit.                                 env BTBA_CHAIN */
    BTBA_OPT_CHAIN_SPARSE_PERIOD  = 11, /* chained launch: 0 (default) an instance's sparse items follow its dense items; R >= 2: every R-th item of an
                                           instance is a sparse one.                                                               env BTBA_CHAIN_PERIOD */
    BTBA_OPT_CHAIN_TIMEOUT_MS     = 12, /* watchdog of the waits inside the chained launch (default 500 ms): see BTBA_ESCHED.        env BTBA_CHAIN_TIMEOUT_MS */
    BTBA_OPT_RELAYOUT             = 14, /* 1: a batch whose correspondences arrive as EntryJ (4 MiB or more, three iterations or more, full frames) is re-laid out to
                                           24-byte records BY ITS FIRST ITERATION'S SWEEP, and the other iterations stream those; 0 (default): every iteration reads
                                           EntryJ.  Same bits; measured a wash at c3 x 32 (the first launch's extra writes cost what the others save).  env BTBA_RELAYOUT */
    BTBA_OPT_CORR_NONTEMPORAL     = 15, /* how the sparse items read the correspondences: -1 (default) with plain loads while the batch's frames + correspondences fit the
                                           memory-side cache (224 MiB of MI355X's 256 MB; env BTBA_LLC_MB), beyond that the instances whose correspondences no longer
                                           fit beside the frames are read with NON-TEMPORAL loads, so that the read-once stream does not evict the frames the dense items
                                           re-read every iteration (c3 x 32: 185 -> 198 k GN it/s; never on object-masked frames); 0: plain loads always; 1: non-temporal
                                           always.  A cache policy: same bits.                                                    env BTBA_CORR_NT */
    BTBA_OPT_COUNT_LIVE           = 13, /* 1: the dense sweep's block-walk workgroups add the number of 8 x 8 pixel blocks they actually walk (the blocks the hull
                                           test could not prove dead) to a counter of the workspace -- setting the option clears it, btba_workspace_live_blocks
                                           reads it.  Measurement aid (bench.py: roofline.executed); one atomic per workgroup while it is on. */
    BTBA_OPT_SOLVE_SMALL          = 16  /* 1 (default): windows of <= 21 frames run the per-instance system solve k_solve_small (round 5: register-resident
                                           multi-wave PCG, frame-sum assembly, sixteen-lane inverse); 0: k_system_solve, the kernel of rounds 1-4, as for larger
                                           windows.  Same sums in another order: results agree to rounding (tests/test_gpu_parity.py).   env BTBA_SOLVE_LEGACY=1 = 0 */
};
BTBA_API int btba_workspace_set_option(btba_workspace *ws, int option, int64_t value);
/* Blocks walked by the dense sweeps enqueued since BTBA_OPT_COUNT_LIVE was last set to 1 (synchronises with the workspace stream). */
BTBA_API int btba_workspace_live_blocks(btba_workspace *ws, uint64_t *blocks);
/* Stream ordering without a host wait, for callers whose producers / consumers run on another HIP stream (PyTorch's
 * current stream, a camera driver's copy stream): _wait_stream makes everything enqueued on the workspace stream AFTER the
 * call wait for what `stream` holds at the time of the call (inputs uploaded or rendered there); _signal_stream makes
 * `stream` wait for what the workspace stream holds (poses / caches / filtered maps produced by the asynchronous entry
 * points).  `stream` = NULL is the legacy default stream.  No-ops when `stream` is the workspace's own stream. */
BTBA_API int btba_workspace_wait_stream(btba_workspace *ws, void *stream);
BTBA_API int btba_workspace_signal_stream(btba_workspace *ws, void *stream);

/* Drop-in for OptimizerGpu::optimizeFrames (LossGPU.cu:53-139).
 *   corres_host       : n_corres EntryJ on the host (any order; pair-major order as produced by
 *                       Bundler::optimizeGPU is used as is, anything else is bucketed by frame pair).
 *   n_match_per_pair  : may be NULL (the reference stores it and never reads it, SBA.cpp:85).  When given -- P = n(n-1)/2
 *                       segment lengths in pair order (0,1) (0,2) ... as Bundler::optimizeGPU builds them -- and adding
 *                       up to n_corres, the array is taken as pair-major and uploaded without a host pass; the device
 *                       verifies every entry against its segment's pair and the call falls back to host bucketing if
 *                       that check fails, so a wrong or inconsistent array costs time, never correctness.
 *   depth_dev[k]      : device float[H*W], metres, 0 = invalid          (Frame.h:73)
 *   normal_dev[k]     : device float4[H*W], xyz unit, w = 0, zeros = invalid (Frame.h:75)
 *   poses_rowmajor    : host float[n_frames*16], camera->model, in/out  (LossGPU.cu:88-97,121-130)
 *   K_rowmajor        : host float[9] full-resolution intrinsics
 *   dense_pairs       : BTBA_PAIRS_EXPLICIT only: n_dense_pairs (target, source) int32 pairs, host.
 * ws may be NULL: like the reference, everything is then allocated and freed inside the call, and the call runs on the
 * legacy NULL stream like the reference does -- depth / normal maps produced by earlier work on the default stream (or by
 * any blocking stream) are ordered before the cache build without the caller doing anything.  A caller that passes NULL
 * only for some calls must not race them against its own non-blocking streams.
 * n_match_per_pair must hold P = n_frames (n_frames - 1) / 2 ints when non-NULL (the host wrappers pass NULL when the
 * caller's vector has another length).
 * Synchronous: poses are valid on return.  Inside the call (round 6) the host is not synchronised before the end: the frame cache is built on the workspace's stream while
 * the EntryJ array and the poses are uploaded on a second, non-blocking stream the workspace owns (it touches the workspace's own buffers only; the solve waits for its event),
 * and every small table crosses through pinned blocks of the workspace -- with ws == NULL those are created and destroyed per call like everything else. */
BTBA_API int btba_optimize_frames(btba_workspace *ws, const btba_params *params,
                         int n_frames, int H, int W, const float *K_rowmajor,
                         const btba_entryj *corres_host, uint32_t n_corres, const int *n_match_per_pair,
                         const float *const *depth_dev, const float *const *normal_dev,
                         const int32_t *dense_pairs, int n_dense_pairs,
                         float *poses_rowmajor, btba_stats *stats);

/* btba_optimize_frames with a PERSISTENT frame cache.  frame_keys[k] is a caller-chosen id
 * of frame k that is stable across calls (the tracker's Frame::_id, Frame.h:61).  A frame whose (key, depth pointer,
 * normal pointer) was cached by an earlier call on the same workspace, with the same H, W, K and image_downscale, is
 * NOT cached again: its compact (z, n) pixels, its valid-pixel list and count stay in a pool slot of the workspace
 * (least-recently-used slots are recycled; pool = max(32, 2 n_frames) slots).  In a tracker only the new frame is
 * built per call, the keyframes were cached when they were new -- the reference re-caches all K frames every call
 * (LossGPU.cu:74-78).  The caller must not modify a frame's device buffers while it is cached under the same key
 * (keyframes are immutable after Frame's constructor, Frame.cpp:107-149); btba_frame_cache_clear drops everything.
 * Results are bit-identical to btba_optimize_frames.  ws must not be NULL; BTBA_FLAG_FLOAT4_CACHE is rejected. */
BTBA_API int btba_optimize_frames_keyed(btba_workspace *ws, const btba_params *params,
                         int n_frames, int H, int W, const float *K_rowmajor,
                         const btba_entryj *corres_host, uint32_t n_corres, const int *n_match_per_pair,
                         const float *const *depth_dev, const float *const *normal_dev, const uint64_t *frame_keys,
                         const int32_t *dense_pairs, int n_dense_pairs,
                         float *poses_rowmajor, btba_stats *stats);
BTBA_API int btba_frame_cache_clear(btba_workspace *ws);
/* Forget ONE cached frame (a tracker dropping a frame whose id it may hand out again, Bundler.cpp:96-104: a failed frame is
 * popped and the next one gets the same id).  Unknown keys are not an error. */
BTBA_API int btba_frame_cache_evict(btba_workspace *ws, uint64_t frame_key);

/* Frame cache build alone (CUDACache::CUDACache + storeFrame, CUDACache.cpp:14-38,76-88).
 * Outputs (device): campos float4[n_frames][Hd*Wd], normals float4[n_frames][Hd*Wd],
 * n_valid int32[n_frames] (may be NULL); intr_out (host) = downscaled (fx, fy, cx, cy).
 * Asynchronous on the workspace stream. */
BTBA_API int btba_build_cache(btba_workspace *ws, int n_frames, int H, int W, const float *K_rowmajor,
                     float image_downscale, const float *const *depth_dev, const float *const *normal_dev,
                     float *campos_dev, float *normals_dev, int32_t *n_valid_dev, float *intr_out);

/* Batched solve over device-resident instances that share n_frames and the cache resolution
 * (the solveBundlingStub seam, SolverBundling.cu:931-1003, for many independent trackers).
 *   campos_dev / normals_dev : float4 [n_instances][n_frames][Hd*Wd]
 *   corr_dev                 : EntryJ [n_instances][corr_stride], each instance pair-major over the
 *                              canonical pair order (0,1),(0,2)..(N-2,N-1)
 *   pair_offsets_dev         : uint32 [n_instances][P+1] segment starts inside the instance's block
 *   max_corr_per_pair        : upper bound of any segment length (sizes the launch)
 *   poses_dev                : float [n_instances][n_frames][16] in/out
 *   dense_pairs (host)       : ordered (target, source) list or NULL for TARGET_LOWER
 *   trace_dev                : NULL or float [n_instances*n_gn_iters*record_floats]
 * Asynchronous on the workspace stream; stats (if non-NULL) are complete after
 * btba_workspace_sync() and a following btba_collect_stats(). */
BTBA_API int btba_solve_batch(btba_workspace *ws, const btba_params *params,
                     int n_instances, int n_frames, int Hd, int Wd, const float *intr,
                     const float *campos_dev, const float *normals_dev,
                     const btba_entryj *corr_dev, int64_t corr_stride,
                     const uint32_t *pair_offsets_dev, uint32_t max_corr_per_pair,
                     const int32_t *dense_pairs, int n_dense_pairs,
                     float *poses_dev, float *trace_dev);
/* SURVEY.md 8(b)'s single-instance cached entry (one tracker, frame cache already built by btba_build_cache): the same
 * solve as btba_solve_batch with n_instances = 1 and corr_stride = n_corr. */
BTBA_API int btba_solve_cached(btba_workspace *ws, const btba_params *params, int n_frames, int Hd, int Wd, const float *intr,
                     const float *campos_dev, const float *normals_dev,
                     const btba_entryj *corr_dev, uint32_t n_corr,
                     const uint32_t *pair_offsets_dev, uint32_t max_corr_per_pair,
                     const int32_t *dense_pairs, int n_dense_pairs,
                     float *poses_dev, float *trace_dev);
BTBA_API int btba_collect_stats(btba_workspace *ws, btba_stats *stats);

BTBA_API void btba_trace_layout_get(int n_frames, int n_dense_pairs, int n_pcg_iters, btba_trace_layout *out);

/* Host helpers (A0/A6 side): bucket arbitrary EntryJ by canonical frame pair.
 * out_sorted (n entries) and out_offsets (P+1) are host buffers.  Returns BTBA_EINVAL if an
 * entry references a frame >= n_frames or has imgIdx_i >= imgIdx_j. Invalid entries are dropped
 * (offsets[P] = number kept). */
BTBA_API int btba_bucket_correspondences(const btba_entryj *in, uint32_t n, int n_frames,
                                btba_entryj *out_sorted, uint32_t *out_offsets);

/* SE(3) seams (convertMatricesToPosesCU / convertPosesToMatricesCU / convertLiePosesToMatricesCU)
 * on device data; n transforms, x = (rot, trans) float[n][6]; any output may be NULL. */
BTBA_API int btba_matrices_to_poses(btba_workspace *ws, int n, const float *T_dev, float *x_dev);
BTBA_API int btba_poses_to_matrices(btba_workspace *ws, int n, const float *x_dev, float *T_dev, float *Tinv_dev);

/* ---- the step before the boundary (what Frame's constructor runs per frame) ---- */
/* Frame::processDepth (src/Frame.cpp:152-180): erode (CUDAImageUtil.cu:676-718) then the mean-gated bilateral
 * filter twice (CUDAImageUtil.cu:735-797), fused in ONE launch.  depth_in_dev / depth_out_dev: device float[H*W]
 * (must not alias).  Defaults of config_ycbineoat.yml:9-16: erode radius 1, diff 0.001, ratio 0.8; filter radius 2,
 * sigma_D 2, sigma_R 100000.  Asynchronous on the workspace stream. */
BTBA_API int btba_process_depth(btba_workspace *ws, int H, int W, const float *depth_in_dev, float *depth_out_dev,
                                int erode_radius, float erode_diff, float erode_ratio,
                                int bf_radius, float sigma_d, float sigma_r);

/* Frame::depthToCloudAndNormals (src/Frame.cpp:182-233): depth -> camera-space points (CUDAImageUtil.cu:310-327)
 * -> normals (computeNormals_Kernel, CUDAImageUtil.cu:342-412), ONE launch.  normals_dev: device float4[H*W]
 * (xyz unit, w = 0, zeros = invalid: the optimiser's input format); xyz_dev: device float4[H*W] or NULL. */
BTBA_API int btba_depth_to_normals(btba_workspace *ws, int H, int W, const float *K_rowmajor,
                                   const float *depth_dev, float *normals_dev, float *xyz_dev);

/* ---- compact frame cache ("ZN") ------------------------------------------------------------------------
 * camPos is a pure function of (full-resolution pixel, depth) -- CUDAImageUtil.cu:310-327 -- so the cache can hold
 * float4 (z, nx, ny, nz) per pixel, 16 B instead of the reference's 32 B (CUDACachedFrame, CUDACacheUtil.h:10-53), and
 * the sweep re-derives camPos with the cache builder's exact fp32 operations: identical results, half the bytes,
 * half the load instructions.  btba_optimize_frames uses it internally (BTBA_FLAG_FLOAT4_CACHE switches back).
 * CONTRACT of the z lane: it is the GATED depth -- 0 wherever the reference's cached camPos is (0, 0, 0, 0), i.e. where
 * the full-resolution depth is below 0.1 m or NaN (CUDAImageUtil.cu:310-327) -- exactly camPos.z of the reference
 * layout.  btba_build_cache_zn and btba_pack_zn produce it; a caller that fills the cache itself must apply the gate. */
BTBA_API int btba_build_cache_zn(btba_workspace *ws, int n_frames, int H, int W, const float *K_rowmajor,
                                 float image_downscale, const float *const *depth_dev, const float *const *normal_dev,
                                 float *zn_dev /* float4[n_frames][Hd*Wd] */, int32_t *n_valid_dev, float *intr_out);
/* float4 camPos + float4 normals (reference layout) -> compact cache.  camPos.xy are dropped and re-derived from z, so
 * the input must have been produced by the standard formula (btba_build_cache or CUDACache). */
BTBA_API int btba_pack_zn(btba_workspace *ws, int64_t n_pixels_total, const float *campos_dev, const float *normals_dev, float *zn_dev);
/* btba_solve_batch on compact caches: zn_dev float4 [n_instances][n_frames][Hd*Wd]; H, W, K_rowmajor describe the
 * FULL-resolution frames the caches were built from (Hd = H / image_downscale, ...). */
BTBA_API int btba_solve_batch_zn(btba_workspace *ws, const btba_params *params, int n_instances, int n_frames,
                                 int H, int W, const float *K_rowmajor, const float *zn_dev,
                                 const btba_entryj *corr_dev, int64_t corr_stride,
                                 const uint32_t *pair_offsets_dev, uint32_t max_corr_per_pair,
                                 const int32_t *dense_pairs, int n_dense_pairs, float *poses_dev, float *trace_dev);

/* Data derived from compact caches ALONE (not from poses or parameters): a caller that keeps its caches across solves builds them
 * once per set of frames and hands them to btba_solve_batch_zn_aux; btba_solve_batch_zn derives what it needs inside every solve
 * (one pass over all frames each: 2 % of a c3 x 32 solve for the block ranges, 6 % of a masked one for the lists);
 * btba_optimize_frames / _keyed keep both with their frame cache.  All pointers are device pointers and optional (NULL).
 *   block_ranges : float2 [n_frames_total][(Hd / 8) * (Wd / 8)] -- depth range [min, max] of the valid pixels of every 8 x 8 block,
 *                  (+inf, -inf) for a block without a valid depth (Hd, Wd multiples of 8).  The pinhole dense sweep uses it to
 *                  drop blocks that provably project outside the target image before touching their pixels (exact: DESIGN.md 4.2).
 *   valid_lists  : uint32 [n_frames_total][Hd * Wd] -- per frame, the ascending list of the pixels that carry a depth,
 *   valid_counts : int32 [n_frames_total] -- and its length; walked instead of all pixels under BTBA_FLAG_COMPACTION. */
typedef struct btba_zn_aux {
    const float *block_ranges;
    const uint32_t *valid_lists;
    const int32_t *valid_counts;
    const float *corr24;              /* the correspondences as 24-byte records (btba_pack_correspondences24): used INSTEAD of corr_dev, with the
                                         same corr_stride / pair_offsets_dev.  NULL: corr_dev (EntryJ) is read */
} btba_zn_aux;
/* Builders (asynchronous on the workspace stream). */
BTBA_API int btba_zn_block_ranges(btba_workspace *ws, int n_frames_total, int Hd, int Wd, const float *zn_dev, float *ranges_dev);
BTBA_API int btba_zn_valid_lists(btba_workspace *ws, int n_frames_total, int Hd, int Wd, const float *zn_dev, uint32_t *lists_dev, int32_t *counts_dev);
/* Device-resident correspondences without their frame indices.  A pair-major array implies (imgIdx_i, imgIdx_j) of every entry through the
 * segment it lies in, so a batch that STAYS on the device (or the keyed pool of btba_optimize_frames_keyed) can drop those 8 of 32 bytes:
 * the sparse sweep streams the array once per Gauss-Newton iteration and is bound by exactly that stream on feature-only windows and on
 * object-masked frames.  24 bytes per entry, kept in groups of 64 entries as three planes of float2 -- (pos_i.x, pos_i.y)[64],
 * (pos_i.z, pos_j.x)[64], (pos_j.y, pos_j.z)[64] -- so that a wave's loads are contiguous: entry E (counted from the start of the array,
 * E = instance * corr_stride + e) has its k-th float2 at float2 index (E / 64) * 192 + k * 64 + E % 64; entry e of the input is entry e of
 * the output (same offsets, same stride, counted in entries).  corr24_dev must hold 24 * 64 * ceil(n_instances * corr_stride / 64) bytes.
 * An invalid entry (imgIdx_i = 0xFFFFFFFF) keeps its place with the bit pattern 0xFFFFFFFF in pos_i.x --
 * consequently a VALID entry whose pos_i.x has that pattern (one particular NaN) is dropped where the reference would propagate the NaN.
 * The wire format at the boundary stays EntryJ (A1).  order_flag_dev (may be NULL): int on the device, ORed with 1 when a valid entry
 * does not carry the pair of its segment (the array was not pair-major).  Asynchronous on the workspace stream. */
BTBA_API int btba_pack_correspondences24(btba_workspace *ws, int n_instances, int n_frames, const btba_entryj *corr_dev, int64_t corr_stride,
                                         const uint32_t *pair_offsets_dev, uint32_t max_corr_per_pair, float *corr24_dev, int32_t *order_flag_dev);
/* btba_solve_batch_zn with the caches' derived data supplied (aux NULL, or any member NULL: as btba_solve_batch_zn). */
BTBA_API int btba_solve_batch_zn_aux(btba_workspace *ws, const btba_params *params, int n_instances, int n_frames,
                                     int H, int W, const float *K_rowmajor, const float *zn_dev, const btba_zn_aux *aux,
                                     const btba_entryj *corr_dev, int64_t corr_stride,
                                     const uint32_t *pair_offsets_dev, uint32_t max_corr_per_pair,
                                     const int32_t *dense_pairs, int n_dense_pairs, float *poses_dev, float *trace_dev);

/* ---- correspondence RANSAC (the step before correspondences enter BA) ------------
 * Replaces ransacMultiPairGPU (src/cuda/cuda_ransac.cu:1228-1323) as called by SiftManager::runRansacMultiPairGPU
 * (FeatureManager.cpp:659-741): for every frame pair, n_trials 3-point rigid hypotheses ptsA -> ptsB, inlier vote with
 * |ptB - pose ptA| <= dist_thres, the inliers of the best trial.
 *   ptsA_host / ptsB_host : float4 (x, y, z, 1) points of ALL pairs back to back (model frame, as the reference uploads
 *                           them); pair p owns n_pts[p] consecutive points.
 *   samples_host          : NULL, or int32 [n_pairs][n_trials][3] explicit sample indices (the rand_list of
 *                           ransacMultiPairKernel, :1105).  NULL: the reference's generator, RESTATED AND UNVERIFIED -- trial t draws
 *                           round(u (n_pts-1)) three times from cuRAND's XORWOW generator after curand_init(seed, t, 0)
 *                           (ransacEstimateModelKernel, :1154-1161; the reference's literal seed is 0 -- pass seed = 0 for its
 *                           triples).  That stream does not depend on the pair, so it is one table of n_trials x 3 uniforms,
 *                           computed on the host from cuRAND's published algorithm (btba_ransac_reference_uniforms below),
 *                           kept in the workspace and read by the vote kernel.  No cuRAND exists on this side: operators, jumps
 *                           and recurrence are pinned against rocRAND's engine, the four seed constants and the uniform
 *                           conversion rest on the published headers (INTEGRATION.md gives the CUDA snippet that settles it).  With BTBA_RANSAC_DRAW_HASH (btba_ransac_pairs_ex)
 *                           the triples come from a counter hash of (seed, pair, trial, draw) instead: same round(u (n-1))
 *                           shape, but every pair gets its own triples.  Trials with repeated / negative / out-of-range
 *                           indices are skipped (:1164-1165).
 *   inlier_ids_out        : int32, same layout as the points: pair p's ascending inlier indices in its first
 *                           n_inliers_out[p] entries.  best_trial_out[p] = -1 when no trial was usable.
 *   best_pose_out         : may be NULL; float [n_pairs][16] row-major 4x4 of the winning 3-point hypothesis.
 *   trial_counts_out / trial_poses_out : may be NULL; int32 [n_pairs][n_trials], float [n_pairs][n_trials][12] (3x4).
 * Best trial = most inliers, lowest trial id among equals (the reference: whichever thread writes last).
 * Deterministic; synchronous (results valid on return).
 * Hypotheses: btba_ransac_pairs uses BTBA_RANSAC_REFERENCE_SVD -- procrustesKernel (cuda_ransac.cu:998-1103) with the reference's
 * APPROXIMATE 3x3 SVD (McAdams et al., UW-Madison TR1690, pasted into cuda_ransac.cu:48-975) restated operation for operation,
 * including its "R is not valid" failure: per-trial poses, inlier counts and the winner equal the reference's on identical sample
 * triples (pinned against the reference's own functions, tests/test_gpu_ransac.py).  BTBA_RANSAC_HORN (btba_ransac_pairs_ex) is the
 * exact Kabsch optimum by Horn's quaternion method instead: never fails, rejects (near-)collinear samples by the eigenvalue gap;
 * on 3-point samples the reference's approximate SVD is more than 4e-3 away from it in ~5 % of the trials.
 * btba_ransac_pairs_ex also takes device_resident = 1: ptsA / ptsB / samples and every output are DEVICE pointers (n_pts stays
 * on the host), nothing but the 4 (n_pairs + 1)-byte offset table crosses PCIe, and the call is asynchronous on the workspace
 * stream (the host-buffer form spends ~60 % of a tracker-size call in its copies). */
enum { BTBA_RANSAC_REFERENCE_SVD = 0, BTBA_RANSAC_HORN = 1,
       BTBA_RANSAC_DRAW_HASH = 0x100 /* ORed into `hypothesis`: counter-hash triples instead of the restated cuRAND stream */ };
BTBA_API int btba_ransac_pairs_ex(btba_workspace *ws, int hypothesis, int device_resident, int n_pairs, const float *ptsA, const float *ptsB,
                                  const int32_t *n_pts, int n_trials, float dist_thres, const int32_t *samples, uint64_t seed,
                                  int32_t *inlier_ids_out, int32_t *n_inliers_out, int32_t *best_trial_out, float *best_pose_out,
                                  int32_t *trial_counts_out, float *trial_poses_out);
BTBA_API int btba_ransac_pairs(btba_workspace *ws, int n_pairs, const float *ptsA_host, const float *ptsB_host, const int32_t *n_pts,
                               int n_trials, float dist_thres, const int32_t *samples_host, uint64_t seed,
                               int32_t *inlier_ids_out, int32_t *n_inliers_out, int32_t *best_trial_out, float *best_pose_out,
                               int32_t *trial_counts_out, float *trial_poses_out);

/* The reference's RANSAC sample stream as numbers (restated from cuRAND's published headers; unverified against a CUDA run): u_out[3 t + k] = the k-th curand_uniform() after curand_init(seed, t, 0) of
 * cuRAND's XORWOW generator, t = 0 .. n_trials-1 (cuda_ransac.cu:1154-1161); trial t of a pair with n points samples
 * round(u * (n - 1)).  Host-only (no GPU, no workspace): Marsaglia's xorwow recurrence, cuRAND's seed scrambling and its
 * 2^67-step subsequence jump as a GF(2) matrix power (bundletrack_amd/csrc/btba_xorwow.hpp).  BTBA_EINVAL on bad arguments. */
BTBA_API int btba_ransac_reference_uniforms(uint64_t seed, int n_trials, float *u_out);

#ifdef __cplusplus
}
#endif
#endif /* BTBA_H_ */
