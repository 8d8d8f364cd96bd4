// btba_api.hip -- host side of libbtba.so: the C ABI declared in include/btba.h.
//
// Mirrors, call for call, what the reference does between Bundler::optimizeGPU and the kernels
// (LossGPU.cu:53-139 -> CUDACache.cpp:76-88 -> SBA.cpp:81-126 -> CUDASolverBundling.cpp:190-280 ->
// SolverBundling.cu:931-1003), re-designed for MI355X: no per-call allocation storm (grow-only
// workspace), no per-iteration host sync (the dense pair list is static), three launches per
// Gauss-Newton iteration, batches of independent instances in one grid.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <string>
#include <map>
#include <memory>
#include <vector>

#include "../../include/btba.h"
#include "btba_kernels.hpp"
#include "btba_solve_small.hpp"
#include "btba_solve_mid.hpp"
#include "btba_image.hpp"
#include "btba_ransac.hpp"
#include "btba_xorwow.hpp"

using namespace btba;

static thread_local int g_last_hip_error = 0;

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) {                         \
            g_last_hip_error = (int)e_;                 \
            return BTBA_EHIP;                           \
        }                                               \
    } while (0)

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return BTBA_OK;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) { g_last_hip_error = (int)e; return BTBA_EHIP; } }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { g_last_hip_error = (int)e; p = nullptr; return e == hipErrorOutOfMemory ? BTBA_ENOMEM : BTBA_EHIP; }
        cap = want;
        return BTBA_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct EventPair { hipEvent_t a, b; int kind; };   // kind 0 dense, 1 sparse, 2 system, 3 solve region, 4 cache

}  // namespace

struct btba_workspace {
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int device = 0;
    DevBuf x, T, Tinv, sparse_part, dense_part, pairsum, dense_pairs, ptrs, big_A, solve_tab;
    DevBuf corr, offsets, poses, campos, normals, nvalid;   // optimize_frames staging
    DevBuf valid_lists, valid_counts;                       // per-frame lists of pixels with a depth (compact cache)
    DevBuf block_ranges;                                    // per (frame, 8 x 8 block) usable depth range: dead-block test of the dense sweep
    DevBuf chain_sync;                                      // chained launch: flags[B] + arrivals[n_gn][B] (zeroed before every launch), optional timeline
    DevBuf chain_trace;
    DevBuf corr24_tmp;                                      // re-layout of a call's EntryJ array written by its first iteration's sparse sweep (BTBA_OPT_RELAYOUT)
    DevBuf live_blocks;                                     // BTBA_OPT_COUNT_LIVE: one uint64 the block-walk workgroups add their walked blocks to
    bool count_live = false;
    int *chain_error = nullptr;                             // pinned host word the chained launch's watchdog raises (checked at every host synchronisation)
    bool chain_failed = false;                              // a watchdog fired on this workspace: chaining stays off from then on
    bool chain_reported = false;                            // ... and an enqueue has already returned BTBA_ESCHED for it (the word itself is cleared only after a sync)
    uint64_t chain_launches = 0;
    // Developer / tuning switches.  Read from the environment ONCE, when the workspace is created (never on the solve path), and settable
    // per workspace through btba_workspace_set_option (include/btba.h: BTBA_OPT_*).  None of them changes what is computed.
    struct Tuning {
        bool dense_order = true;       // BTBA_OPT_DENSE_ORDER   (env BTBA_NO_DENSE_ORDER=1 turns it off): dense pairs worked off heaviest first
        bool tile_major = true;        // BTBA_OPT_TILE_MAJOR    (env BTBA_PAIR_MAJOR=1 turns it off): (band, pair) instead of (pair, band) work order
        bool block_walk = true;        // BTBA_OPT_BLOCK_WALK    (env BTBA_NO_BLOCK_WALK=1): waves walk 8 x 8 blocks instead of 64 x 1 strips
        bool block_skip = true;        // BTBA_OPT_BLOCK_SKIP    (env BTBA_NO_BLOCK_SKIP=1): provably dead blocks are not walked
        int sparse_tail_256 = -1;      // BTBA_OPT_SPARSE_TAIL   (env BTBA_SPARSE_TAIL): share (x / 256) of the sparse items that close the fused launch; -1 = the library's choice
        bool big_assembly = true;      // BTBA_OPT_BIG_ASSEMBLY  (env BTBA_NO_BIG_ASSEMBLY=1): many-workgroup reduction / assembly from 24 frames on
        int overlap_groups = 2;        // BTBA_OPT_OVERLAP_GROUPS (env BTBA_GROUPS): instance groups of BTBA_FLAG_OVERLAP
        bool overlap_equal_prio = false;   // BTBA_OPT_OVERLAP_EQUAL_PRIO (env BTBA_GROUP_PRIO=e...)
        size_t keyed_corr_min_bytes = (size_t)1 << 20;   // BTBA_OPT_KEYED_CORR_MIN_BYTES (env of the same name): below it the keyed correspondence cache is not used
        int chain = 0;                 // BTBA_OPT_CHAIN         (env BTBA_CHAIN): 1 = all Gauss-Newton iterations of a batch in ONE launch (k_chain) whenever the launch supports the solve; 0 (default) / -1 = the plain schedule
        int corr_nt = -1;              // BTBA_OPT_CORR_NONTEMPORAL (env BTBA_CORR_NT): non-temporal correspondence loads  1 always, 0 never, -1 (default) the library's choice (corr_nt_auto)
        int corr_nt_partial = 1;       // env BTBA_CORR_NT_PARTIAL=0 (developer): all instances stream non-temporally once the batch exceeds the cache, not only those that do not fit
        long long last_level_cache = 224ll << 20;   // env BTBA_LLC_MB: what of the 256 MB memory-side cache a batch's frames + correspondences may fill before the stream is read non-temporally
        bool relayout = false;         // BTBA_OPT_RELAYOUT (env BTBA_RELAYOUT=1 turns it on): a batch given as EntryJ is re-laid out to 24-byte records by its first iteration's sweep
        int chain_group = 1;           // env BTBA_CHAIN_GROUP (developer A/B): instances per group of the chained launch's sequence (ChainDims::group)
        int chain_sparse_period = 0;   // BTBA_OPT_CHAIN_SPARSE_PERIOD (env BTBA_CHAIN_PERIOD): 0 = an instance's sparse items follow its dense items, R >= 2 = every R-th item is a sparse one
        int chain_timeout_ms = 500;    // BTBA_OPT_CHAIN_TIMEOUT_MS (env BTBA_CHAIN_TIMEOUT_MS): watchdog of the waits inside the chained launch
        int chain_solve_prio = 0;      // env BTBA_CHAIN_SOLVE_PRIO (developer A/B): s_setprio of the solve items' waves
        int chain_debug_skip = 0;      // env BTBA_CHAIN_DEBUG_SKIP (developer TIMING experiments, wrong results): ChainDims::debug_skip
        bool solve_small = true;       // BTBA_OPT_SOLVE_SMALL (env BTBA_SOLVE_LEGACY=1 turns it off): k_solve_small for windows of <= 21 frames
        int prepare_keep_T = 0;        // env BTBA_PREPARE_KEEP_T (developer / experiment builds): a solve's incoming matrices are its first iterate's T as they are (k_prepare)
        int debug_lds_pad = 0;         // env BTBA_DEBUG_LDS_PAD (developer): extra dynamic LDS bytes per sweep workgroup -- what a larger LDS footprint costs the fused sweep
        std::string chain_trace_file;  // env BTBA_CHAIN_TRACE_FILE (developer, scripts/chain_trace.py): every chained solve synchronises and dumps its workgroup timeline there
    } tune;
    std::vector<int32_t> dense_pairs_host;                  // what dense_pairs currently holds
    int dense_pairs_frames = -1;
    size_t dense_work_offset = 0;                           // ints into dense_pairs: the fused sweep's work table
    int work_formula = 0;                                   // SolveDims::work_formula of that table
    int solve_tab_frames = -1;                              // window size solve_tab was built for
    std::vector<EventPair> events;                          // pending timed regions
    std::vector<hipEvent_t> event_pool;
    btba_stats stats{};
    bool lds_attr_set = false, small_attr_set = false, mid_attr_set = false;
    int n_cus = 0;                     // compute units of the workspace's device (256 = all eight XCDs of an MI355X in SPX mode: what k_chain's item -> XCD mapping assumes)
    bool always_time_region = false;   // optimize_frames: ms_solve is part of its stats contract
    static constexpr int kMaxGroups = 8;
    uint64_t solves_enqueued = 0;      // rotates the sampled iteration of BTBA_FLAG_TIME_SAMPLED
    hipStream_t aux_streams[kMaxGroups - 1] = {};  // groups 1 .. G-1 of a batch run here (software pipelining across instances)
    hipEvent_t ev_fork = nullptr, ev_join[kMaxGroups - 1] = {}, ev_order = nullptr;
    // optimize_frames (round 6): the EntryJ / pose upload runs on a stream of its own while the frame cache is built on `stream`; small tables the cache build and
    // the solve need (pointer tables, slot maps, valid counts) go through ONE pinned staging block, so that no call has to synchronise just to keep a local alive
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copy = nullptr, ev_cache = nullptr;
    void *pin = nullptr; size_t pin_cap = 0;
    void *pin_io = nullptr; size_t pin_io_cap = 0;          // pinned: [poses out | poses in | pair offsets] of one optimize_frames call (small pageable copies cost ~10 us of host time each)
    int pin_io_ensure(size_t bytes)
    {
        if (bytes <= pin_io_cap) return BTBA_OK;
        if (pin_io) { (void)hipHostFree(pin_io); pin_io = nullptr; pin_io_cap = 0; }
        const size_t want = bytes + bytes / 2 + 4096;
        hipError_t e = hipHostMalloc(&pin_io, want, hipHostMallocDefault);
        if (e != hipSuccess) { g_last_hip_error = (int)e; pin_io = nullptr; return e == hipErrorOutOfMemory ? BTBA_ENOMEM : BTBA_EHIP; }
        pin_io_cap = want;
        return BTBA_OK;
    }
    int pin_ensure(size_t bytes)
    {
        if (bytes <= pin_cap) return BTBA_OK;
        if (pin) { (void)hipHostFree(pin); pin = nullptr; pin_cap = 0; }
        const size_t want = bytes + bytes / 2 + 4096;
        hipError_t e = hipHostMalloc(&pin, want, hipHostMallocDefault);
        if (e != hipSuccess) { g_last_hip_error = (int)e; pin = nullptr; return e == hipErrorOutOfMemory ? BTBA_ENOMEM : BTBA_EHIP; }
        pin_cap = want;
        return BTBA_OK;
    }
    struct PendingSlot { int slot; uint64_t key; const float *depth, *normal; };
    std::vector<PendingSlot> pool_pending;                  // frames cached by the call in flight: committed (live, n_valid) once their counts have come back

    // persistent frame cache (btba_optimize_frames_keyed): compact (z, n) frames, their valid-pixel lists and counts
    // live in pool slots that survive across calls; a keyframe is cached once, not once per BA call.
    struct FrameSlot { uint64_t key = 0; const float *depth = nullptr, *normal = nullptr; uint64_t stamp = 0; bool live = false; int32_t n_valid = 0; };
    DevBuf pool_zn, pool_lists, pool_counts, pool_nvalid, pool_map, pool_ranges;
    size_t pool_map_offset = 0;                             // bytes into pool_map at which the window's frame -> slot map starts (behind the call's pointer table)
    // keyed correspondence cache (BTBA_FLAG_KEYED_CORR): the EntryJ segment of a frame PAIR stays on the device under the pair's two
    // frame keys; a sliding window then uploads only the new frame's K - 1 segments
    struct CorrSeg { uint32_t off = 0, count = 0; };
    DevBuf corr_pool, corr_desc, corr_stage_dev, corr_lens;   // pool of 24-byte correspondences; staging of a call's fresh EntryJ segments; the window's segment lengths
    std::map<std::pair<uint64_t, uint64_t>, CorrSeg> corr_index;
    size_t corr_pool_used = 0;                              // in entries
    void *corr_stage = nullptr; size_t corr_stage_cap = 0;  // pinned host staging of the segments uploaded by one call
    DevBuf ransac;                                          // btba_ransac_pairs staging (points, samples, per-trial poses and counts, results)
    DevBuf ransac_u;                                        // the reference's sample stream: n_trials x 3 uniforms (btba_xorwow.hpp), kept per (seed, n_trials)
    std::vector<float> ransac_u_host;
    uint64_t ransac_u_seed = 0;
    std::vector<FrameSlot> pool_slots;
    int pool_H = 0, pool_W = 0, pool_npix = 0;
    float pool_downscale = 0.0f, pool_K[9] = {0};
    uint64_t pool_stamp = 0;
    uint64_t pool_hits = 0, pool_misses = 0;

    hipEvent_t get_event()
    {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};

// A workspace belongs to the device that was current when it was created.  A process that drives several GPUs from one thread
// ("one process looping hipSetDevice") may call in with another device current: every entry point that takes a
// workspace switches to the workspace's device for the duration of the call and back afterwards.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const btba_workspace *ws)
    {
        int cur = -1;
        if (ws && hipGetDevice(&cur) == hipSuccess && cur != ws->device && hipSetDevice(ws->device) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

extern "C" {

void btba_params_default(btba_params *p)
{
    if (!p) return;
    p->n_gn_iters = 7;
    p->n_pcg_iters = 5;
    p->robust_delta = 0.005f;
    p->dense_dist_thresh = 0.02f;
    p->dense_normal_thresh = (float)std::cos(45.0 / 180.0 * M_PI);
    p->depth_min = 0.1f;
    p->depth_max = 9999.0f;
    p->weight_sparse = 1.0f;
    p->weight_dense_depth = 1.0f;
    p->image_downscale = 4.0f;
    p->pair_policy = BTBA_PAIRS_TARGET_LOWER;
    p->dense_tiles = 0;
    p->sparse_chunks = 0;
    p->flags = 0;
    p->reduction_mode = BTBA_REDUCE_DETERMINISTIC;
    p->weights_sparse_per_iter = nullptr;
    p->weights_dense_per_iter = nullptr;
    p->n_weights_per_iter = 0;
}

const char *btba_strerror(int status)
{
    switch (status) {
    case BTBA_OK: return "ok";
    case BTBA_EINVAL: return "invalid argument";
    case BTBA_EHIP: return "HIP runtime error (see btba_last_hip_error)";
    case BTBA_ENUMERIC: return "non-finite value in the output poses";
    case BTBA_ENOMEM: return "out of device memory";
    case BTBA_ESCHED: return "a wait inside the chained launch timed out (workgroups did not start in grid order): the solve's poses are invalid; the workspace now solves unchained";
    default: return "unknown status";
    }
}

int btba_last_hip_error(void) { return g_last_hip_error; }
int btba_version(void) { return BTBA_VERSION; }

static int workspace_create(btba_workspace **out, void *stream, bool use_given);

int btba_workspace_create(btba_workspace **out, void *stream) { return workspace_create(out, stream, stream != nullptr); }
int btba_workspace_create_on_stream(btba_workspace **out, void *stream) { return workspace_create(out, stream, true); }

static int workspace_create(btba_workspace **out, void *stream, bool use_given)
{
    if (!out) return BTBA_EINVAL;
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { g_last_hip_error = (int)hipErrorNoDevice; return BTBA_EHIP; }
    btba_workspace *ws = new (std::nothrow) btba_workspace();
    if (!ws) return BTBA_ENOMEM;
    if (hipGetDevice(&ws->device) != hipSuccess) { delete ws; return BTBA_EHIP; }
    if (hipDeviceGetAttribute(&ws->n_cus, hipDeviceAttributeMultiprocessorCount, ws->device) != hipSuccess) ws->n_cus = 0;
    {   // developer switches from the environment: here and nowhere else
        auto on = [](const char *name) { const char *e = std::getenv(name); return e && e[0] && e[0] != '0'; };
        btba_workspace::Tuning &t = ws->tune;
        t.dense_order = !on("BTBA_NO_DENSE_ORDER");
        t.tile_major = !on("BTBA_PAIR_MAJOR");
        t.block_walk = !on("BTBA_NO_BLOCK_WALK");
        t.block_skip = !on("BTBA_NO_BLOCK_SKIP");
        t.big_assembly = !on("BTBA_NO_BIG_ASSEMBLY");
        if (const char *e = std::getenv("BTBA_SPARSE_TAIL")) t.sparse_tail_256 = std::max(0, std::min(256, std::atoi(e)));
        if (const char *e = std::getenv("BTBA_GROUPS")) t.overlap_groups = std::atoi(e);
        if (const char *e = std::getenv("BTBA_GROUP_PRIO")) t.overlap_equal_prio = e[0] == 'e';
        if (const char *e = std::getenv("BTBA_KEYED_CORR_MIN_BYTES")) t.keyed_corr_min_bytes = (size_t)std::strtoull(e, nullptr, 10);
        if (const char *e = std::getenv("BTBA_CHAIN")) t.chain = std::max(-1, std::min(1, std::atoi(e)));
        if (const char *e = std::getenv("BTBA_CHAIN_PERIOD")) t.chain_sparse_period = std::max(0, std::atoi(e));
        t.relayout = on("BTBA_RELAYOUT");
        t.solve_small = !on("BTBA_SOLVE_LEGACY");
        if (const char *e = std::getenv("BTBA_CORR_NT")) t.corr_nt = std::atoi(e);
        if (const char *e = std::getenv("BTBA_LLC_MB")) t.last_level_cache = (long long)std::atoll(e) << 20;
        if (const char *e = std::getenv("BTBA_CORR_NT_PARTIAL")) t.corr_nt_partial = std::atoi(e);
        if (const char *e = std::getenv("BTBA_CHAIN_TIMEOUT_MS")) t.chain_timeout_ms = std::max(1, std::atoi(e));
#if defined(BTBA_DEV_EXPERIMENTS) || defined(BTBA_REFERENCE_ORDER)
        t.prepare_keep_T = on("BTBA_PREPARE_KEEP_T") ? 1 : 0;
#endif
#ifdef BTBA_DEV_EXPERIMENTS      // developer builds only (scripts/chain_trace.py and the timing experiments of profiles/r04): these change schedules in ways a product build never does
        if (const char *e = std::getenv("BTBA_CHAIN_GROUP")) t.chain_group = std::max(1, std::atoi(e));
        if (const char *e = std::getenv("BTBA_CHAIN_TRACE_FILE")) t.chain_trace_file = e;
        if (const char *e = std::getenv("BTBA_CHAIN_SOLVE_PRIO")) t.chain_solve_prio = std::max(0, std::min(3, std::atoi(e)));
        if (const char *e = std::getenv("BTBA_CHAIN_DEBUG_SKIP")) t.chain_debug_skip = std::atoi(e);
        if (const char *e = std::getenv("BTBA_DEBUG_LDS_PAD")) t.debug_lds_pad = std::max(0, std::min(60000, std::atoi(e)));
#endif
    }
    if (use_given) {
        ws->stream = reinterpret_cast<hipStream_t>(stream);       // may be the NULL stream
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { g_last_hip_error = (int)e; delete ws; return BTBA_EHIP; }
        ws->owns_stream = true;
    }
    *out = ws;
    return BTBA_OK;
}

void btba_workspace_destroy(btba_workspace *ws)
{
    DeviceGuard device_guard(ws);
    if (!ws) return;
    (void)hipStreamSynchronize(ws->stream);
    for (auto &ep : ws->events) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    for (auto e : ws->event_pool) (void)hipEventDestroy(e);
    DevBuf *bufs[] = { &ws->x, &ws->T, &ws->Tinv, &ws->sparse_part, &ws->dense_part, &ws->pairsum, &ws->dense_pairs, &ws->ptrs, &ws->big_A, &ws->solve_tab,
                       &ws->corr, &ws->offsets, &ws->poses, &ws->campos, &ws->normals, &ws->nvalid, &ws->valid_lists, &ws->valid_counts, &ws->block_ranges,
                       &ws->chain_sync, &ws->chain_trace, &ws->live_blocks, &ws->corr24_tmp, &ws->pool_zn, &ws->pool_lists, &ws->pool_counts, &ws->pool_nvalid, &ws->pool_map, &ws->pool_ranges, &ws->ransac, &ws->ransac_u, &ws->corr_pool, &ws->corr_desc, &ws->corr_stage_dev, &ws->corr_lens };
    for (auto b : bufs) b->release();
    if (ws->corr_stage) (void)hipHostFree(ws->corr_stage);
    if (ws->chain_error) (void)hipHostFree(ws->chain_error);
    for (auto st : ws->aux_streams) if (st) (void)hipStreamDestroy(st);
    if (ws->copy_stream) (void)hipStreamDestroy(ws->copy_stream);
    if (ws->ev_copy) (void)hipEventDestroy(ws->ev_copy);
    if (ws->ev_cache) (void)hipEventDestroy(ws->ev_cache);
    if (ws->pin) (void)hipHostFree(ws->pin);
    if (ws->pin_io) (void)hipHostFree(ws->pin_io);
    if (ws->ev_fork) (void)hipEventDestroy(ws->ev_fork);
    for (auto e : ws->ev_join) if (e) (void)hipEventDestroy(e);
    if (ws->ev_order) (void)hipEventDestroy(ws->ev_order);
    if (ws->owns_stream) (void)hipStreamDestroy(ws->stream);
    delete ws;
}

int btba_workspace_set_option(btba_workspace *ws, int option, int64_t value)
{
    DeviceGuard device_guard(ws);      // (BTBA_OPT_COUNT_LIVE allocates, BTBA_OPT_OVERLAP_EQUAL_PRIO destroys streams: on the workspace's device, whatever is current)
    if (!ws) return BTBA_EINVAL;
    btba_workspace::Tuning &t = ws->tune;
    switch (option) {
    case BTBA_OPT_DENSE_ORDER: t.dense_order = value != 0; ws->dense_pairs_frames = -1; break;      // the work table is rebuilt
    case BTBA_OPT_TILE_MAJOR: t.tile_major = value != 0; break;
    case BTBA_OPT_BLOCK_WALK: t.block_walk = value != 0; break;
    case BTBA_OPT_BLOCK_SKIP: t.block_skip = value != 0; break;
    case BTBA_OPT_BIG_ASSEMBLY: t.big_assembly = value != 0; break;
    case BTBA_OPT_SPARSE_TAIL: if (value < -1 || value > 256) return BTBA_EINVAL; t.sparse_tail_256 = (int)value; break;
    case BTBA_OPT_OVERLAP_GROUPS: if (value < 1 || value > btba_workspace::kMaxGroups) return BTBA_EINVAL; t.overlap_groups = (int)value; break;
    case BTBA_OPT_OVERLAP_EQUAL_PRIO:
        if (t.overlap_equal_prio != (value != 0)) {        // the groups' streams carry their priority from creation: drop them, the next BTBA_FLAG_OVERLAP solve makes new ones
            for (auto &st : ws->aux_streams) if (st) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipStreamDestroy(st)); st = nullptr; }
        }
        t.overlap_equal_prio = value != 0;
        break;
    case BTBA_OPT_KEYED_CORR_MIN_BYTES: if (value < 0) return BTBA_EINVAL; t.keyed_corr_min_bytes = (size_t)value; break;
    case BTBA_OPT_CHAIN: if (value < -1 || value > 1) return BTBA_EINVAL; t.chain = (int)value; break;
    case BTBA_OPT_CHAIN_SPARSE_PERIOD: if (value < 0 || value == 1 || value > 64) return BTBA_EINVAL; t.chain_sparse_period = (int)value; break;
    case BTBA_OPT_CHAIN_TIMEOUT_MS: if (value < 1 || value > 60000) return BTBA_EINVAL; t.chain_timeout_ms = (int)value; break;
    case BTBA_OPT_RELAYOUT: t.relayout = value != 0; break;
    case BTBA_OPT_SOLVE_SMALL: t.solve_small = value != 0; break;
    case BTBA_OPT_CORR_NONTEMPORAL: if (value < -1 || value > 1) return BTBA_EINVAL; t.corr_nt = (int)value; break;
    case BTBA_OPT_COUNT_LIVE:
        ws->count_live = value != 0;
        if (ws->count_live) {
            int rc = ws->live_blocks.ensure(8 * sizeof(unsigned long long));       // [0] walked blocks; [1 .. 8): the lane census of developer builds (-DBTBA_CENSUS)
            if (rc) return rc;
            HIP_TRY(hipMemsetAsync(ws->live_blocks.p, 0, 8 * sizeof(unsigned long long), ws->stream));
        }
        break;
    case 1000:      // not part of the ABI.  64 = the watchdog's self-test (solve items never publish: the launch runs into the watchdog, the solve is REPORTED failed,
                    // tests/test_gpu_chain.py); the timing experiments (other bits: sweep items that skip their waits, solve items that do nothing -- wrong poses
                    // behind BTBA_OK) exist in developer builds only
#ifdef BTBA_DEV_EXPERIMENTS
        t.chain_debug_skip = (int)value; break;
#else
        if (value != 0 && value != 64) return BTBA_EINVAL;
        t.chain_debug_skip = (int)value; break;
#endif
    default: return BTBA_EINVAL;
    }
    return BTBA_OK;
}

// After a host synchronisation: did a wait inside a chained launch run into its watchdog?  The poses of that solve are then unusable; the
// call reports BTBA_ESCHED once and the workspace solves unchained from then on.
static int chain_check(btba_workspace *ws)
{
    if (!ws->chain_error || !*ws->chain_error) return BTBA_OK;
    *ws->chain_error = 0;                                   // the stream has just been synchronised: the poison kernel of the failed launch has run
    ws->chain_failed = true;
    const bool already = ws->chain_reported;
    ws->chain_reported = false;
    return already ? BTBA_OK : BTBA_ESCHED;
}

int btba_workspace_live_blocks(btba_workspace *ws, uint64_t *blocks)
{
    DeviceGuard device_guard(ws);
    if (!ws || !blocks) return BTBA_EINVAL;
    *blocks = 0;
    if (!ws->live_blocks.p) return BTBA_OK;
    HIP_TRY(hipStreamSynchronize(ws->stream));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, ws->live_blocks.p, sizeof v, hipMemcpyDeviceToHost));
    *blocks = v;
    return BTBA_OK;
}

#ifdef BTBA_CENSUS
// Developer builds only (scripts/sweep_census.py): the lane census of the dense block walk since BTBA_OPT_COUNT_LIVE was set --
// out[0] walked blocks (= wave trips), [1] lanes with a usable source depth, [2] lanes whose projection lands in the target image (`valid`),
// [3] wave trips that end at ballot(valid) == 0, [4] lanes in the trips that go on (64 per trip), [5] lanes accepted after the tap tests,
// [6] lanes valid but rejected by the target depth range, [7] lanes valid, depth fine, rejected by the normal / distance tests.
extern "C" BTBA_API int btba_dev_census(btba_workspace *ws, uint64_t *out)
{
    DeviceGuard device_guard(ws);
    if (!ws || !out || !ws->live_blocks.p) return BTBA_EINVAL;
    HIP_TRY(hipStreamSynchronize(ws->stream));
    HIP_TRY(hipMemcpy(out, ws->live_blocks.p, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return BTBA_OK;
}
#endif

int btba_workspace_sync(btba_workspace *ws)
{
    DeviceGuard device_guard(ws);
    if (!ws) return BTBA_EINVAL;
    HIP_TRY(hipStreamSynchronize(ws->stream));
    return chain_check(ws);
}

static int order_streams(btba_workspace *ws, hipStream_t from, hipStream_t to)
{
    if (from == to) return BTBA_OK;
    if (!ws->ev_order) HIP_TRY(hipEventCreateWithFlags(&ws->ev_order, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ws->ev_order, from));
    HIP_TRY(hipStreamWaitEvent(to, ws->ev_order, 0));
    return BTBA_OK;
}

int btba_workspace_wait_stream(btba_workspace *ws, void *stream)
{
    DeviceGuard device_guard(ws);
    if (!ws) return BTBA_EINVAL;
    return order_streams(ws, static_cast<hipStream_t>(stream), ws->stream);
}

int btba_workspace_signal_stream(btba_workspace *ws, void *stream)
{
    DeviceGuard device_guard(ws);
    if (!ws) return BTBA_EINVAL;
    return order_streams(ws, ws->stream, static_cast<hipStream_t>(stream));
}

void btba_trace_layout_get(int n_frames, int n_dense_pairs, int n_pcg_iters, btba_trace_layout *L)
{
    if (!L) return;
    const int64_t N = n_frames, n = 6 * N;
    int64_t o = 0;
    L->off_x = o; o += N * 6;
    L->off_T = o; o += N * 16;
    L->off_rhs = o; o += N * 6;
    L->off_precond = o; o += N * 6;
    L->off_pcg = o; o += (int64_t)n_pcg_iters * 4;
    L->off_delta = o; o += N * 6;
    L->off_dense_pair = o; o += (int64_t)n_dense_pairs * kDenseVals;
    L->off_A = o; o += n * n;
    L->off_clk = o; o += 8;
    L->record_floats = o;
}

static inline int pair_index(int n_frames, uint32_t i, uint32_t j) { return (int)(i * n_frames - i * (i + 1) / 2 + (j - i - 1)); }

// One validating pass: per-pair counts -> offsets, and whether the input is already pair-major with nothing to drop
// (the only order Bundler::optimizeGPU produces, Bundler.cpp:298-323) so that callers can skip the scatter.
static int count_correspondences(const btba_entryj *in, uint32_t n, int n_frames, uint32_t *out_offsets, bool *already_bucketed)
{
    const int P = n_frames * (n_frames - 1) / 2;
    std::vector<uint32_t> cnt(P + 1, 0);
    bool ordered = true;
    int prev = 0;
    for (uint32_t e = 0; e < n; e++) {
        const btba_entryj &c = in[e];
        if (c.imgIdx_i == 0xFFFFFFFFu) { ordered = false; continue; }
        if (c.imgIdx_i >= (uint32_t)n_frames || c.imgIdx_j >= (uint32_t)n_frames || c.imgIdx_i >= c.imgIdx_j) return BTBA_EINVAL;
        const int p = pair_index(n_frames, c.imgIdx_i, c.imgIdx_j);
        ordered = ordered && (p >= prev);
        prev = p;
        cnt[p + 1]++;
    }
    out_offsets[0] = 0;
    for (int p = 0; p < P; p++) out_offsets[p + 1] = out_offsets[p] + cnt[p + 1];
    if (already_bucketed) *already_bucketed = ordered;
    return BTBA_OK;
}

static void scatter_correspondences(const btba_entryj *in, uint32_t n, int n_frames, const uint32_t *offsets, btba_entryj *out_sorted)
{
    const int P = n_frames * (n_frames - 1) / 2;
    std::vector<uint32_t> cur(offsets, offsets + P);
    for (uint32_t e = 0; e < n; e++) {                       // stable: keeps the caller's order inside a pair
        const btba_entryj &c = in[e];
        if (c.imgIdx_i == 0xFFFFFFFFu) continue;
        out_sorted[cur[pair_index(n_frames, c.imgIdx_i, c.imgIdx_j)]++] = c;
    }
}

int btba_bucket_correspondences(const btba_entryj *in, uint32_t n, int n_frames, btba_entryj *out_sorted, uint32_t *out_offsets)
{
    if ((!in && n) || n_frames < 2 || !out_sorted || !out_offsets) return BTBA_EINVAL;
    if (int rc = count_correspondences(in, n, n_frames, out_offsets, nullptr)) return rc;
    scatter_correspondences(in, n, n_frames, out_offsets, out_sorted);
    return BTBA_OK;
}

}  // extern "C"

// ---- internal: enqueue one batched solve on ws->stream ------------------------------------------
static int time_begin(btba_workspace *ws, bool on, int kind, size_t *slot, hipStream_t st = nullptr)
{
    if (!st) st = ws->stream;
    *slot = (size_t)-1;
    if (!on) return BTBA_OK;
    EventPair ep{ ws->get_event(), ws->get_event(), kind };
    if (!ep.a || !ep.b) return BTBA_EHIP;
    HIP_TRY(hipEventRecord(ep.a, st));
    ws->events.push_back(ep);
    *slot = ws->events.size() - 1;
    return BTBA_OK;
}
static int time_end(btba_workspace *ws, size_t slot, hipStream_t st = nullptr)
{
    if (!st) st = ws->stream;
    if (slot == (size_t)-1) return BTBA_OK;
    HIP_TRY(hipEventRecord(ws->events[slot].b, st));
    return BTBA_OK;
}

static int pick_chunks(const btba_params *prm, int B, int P, uint32_t max_corr_per_pair, bool with_dense)
{
    if (prm->sparse_chunks > 0) return prm->sparse_chunks;
    if (max_corr_per_pair == 0) return 1;
    const long blocks = (long)B * P;
    if (with_dense) {
        // The sparse items ride in the dense items' launch (k_fused_sweeps needs >= 64 of each): as FEW partials per sum as keep it fused -- every extra chunk is
        // another record per pair for the system solve to fetch and add, and the sparse items are not what the launch waits for.  Round 5, one window with
        // k_solve_small (scripts/dev/single_window_tiles.py, ms per solve for 1 / 2 / 4 chunks): c3 masked at one tile 0.126 / 0.138 / 0.145, c3 full frames at
        // eight tiles 0.1765 / 0.1773 / 0.1793; a 10-frame window (45 pairs) needs two chunks to stay fused: 0.116 against 0.142 unfused.
        const int cap = std::min(16, (int)((max_corr_per_pair + 2 * kBlock - 1) / (2 * kBlock)));
        const int want = (int)((64 + blocks - 1) / blocks);
        if (want <= cap) return std::max(1, want);
        // (too few pairs to fuse even at the cap: the sparse sweep is a launch of its own -- the rule below)
    }
    int want = (int)((512 + blocks - 1) / blocks);
    // a chunk is at least one full trip of its workgroup (two entries per lane): a single c3 window measured 0.2191 / 0.2145 / 0.2143 / 0.2283 ms
    // per solve for 5 / 4 / 2 / 8 chunks of its 2 000-entry segments (scripts/latency_tiles.py, r03 call 58)
    const int cap = (int)((max_corr_per_pair + 2 * kBlock - 1) / (2 * kBlock));
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    if (want > 16) want = 16;
    return want;
}
static int pick_tiles(const btba_params *prm, int B, int Pd, int npix, bool lists, int Wd, int Hd)
{
    if (prm->dense_tiles > 0) return prm->dense_tiles;
    if (Pd == 0) return 1;
    // Workgroups per dense frame pair.  Every workgroup pays its set-up (item, poses, ray tables, hull test: ~7 of a two-tile item's 35 us at c3), so a
    // launch wants the FEWEST tiles that still fill the chip: about one dense item per resident workgroup slot (1 536).  Measured with the fused launch
    // of round 4 (hull-culled block walk, sparse items closing the launch), c3 (105 pairs per instance), ms per step for 1 / 2 / 3 / 4 / 5 / 8 tiles,
    // gpurun_out/r04_17 ... r04_19 (profiles/r04/tile_sweep.json):
    //   B = 1   0.521 / 0.317 / 0.281 / 0.267 / 0.261 / 0.250        B = 2   0.504 / 0.344 / 0.315 / 0.279 / 0.295 / 0.297
    //   B = 4   0.528 / 0.368 / 0.375 / 0.352 / 0.391 / 0.392        B = 8   0.519 / 0.457 / 0.468 / 0.487 / 0.512 / 0.571
    //   B = 16  0.688 / 0.698 / -     / 0.771                        B = 32  1.247 / 1.306        B = 64  2.351 / 2.428
    // (round 2's table, taken before the dead blocks were culled and the sparse items moved to the end of the launch, had 2 tiles from 2 048
    // pair-instances on and 4 from 512: "one tile per pair 2.07 against 1.52 ms, the drain dominates" no longer holds -- the short sparse items
    // fill the drain of the long dense ones.)
    const long blocks = (long)B * Pd;
    // object-masked frames walked through their valid-pixel lists (~5 % of the image): 1 / 2 / 3 / 5 tiles  B = 1: 0.219 / 0.216 / 0.215 / 0.221,
    // B = 8: 0.274 / 0.288 / 0.306 / 0.348,  B = 32: 0.494 / 0.554 / 0.624 / 0.785
    // (round 5, with k_solve_small: one masked window 0.1263 / 0.1362 / 0.1331 / 0.141 ms per solve for 1 / 2 / 3 / 4 tiles at one chunk, two windows 0.1355 / 0.1493 /
    // 0.1507 -- one tile from the point where the dense items alone keep the launch fused (>= 64 of them); a 10-frame window's 45 pairs: 3 tiles, 0.116 against 0.145)
    if (lists) return blocks >= 64 ? 1 : 3;
    int want = blocks >= 1536 ? 1 : blocks >= 768 ? 2 : blocks >= 192 ? 4 : 8;
    const int cap = (npix + kBlock - 1) / kBlock;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    // the block walk splits the image by rows of 8 x 8 blocks: tiles beyond ceil(rows / rows-per-tile) would be empty workgroups and
    // empty partials (160 x 120: 15 block rows, 10 tiles -> 2 rows per tile -> 8 tiles; a single instance: 0.301 -> 0.269 ms per solve)
    if (Wd % 8 == 0 && Hd % 8 == 0) {
        // ... and a band must not hold more than 1 024 blocks (4 x 256 lanes test a band's blocks: solve_enqueue drops the hull-culled walk for row strips
        // beyond that).  The one-tile policy above was measured on 160 x 120 caches (300 blocks); a 320 x 240 cache (image_downscale 2: 1 200 blocks)
        // in a chip-filling batch gets the two tiles that keep the walk
        const int bw = Wd / 8, bh = Hd / 8;
        const int min_tiles = (bw * bh + 1023) / 1024;
        if (want < min_tiles && bw <= 1024) want = std::min(min_tiles, bh);
        const int rows_per = (bh + want - 1) / want;
        want = (bh + rows_per - 1) / rows_per;
        while (want < bh && (long)((bh + want - 1) / want) * bw > 1024 && bw <= 1024) want++;      // (rows per tile round up)
    }
    return want;
}

static void scaled_intrinsics(int H, int W, int Hd, int Wd, const float *K, float intr[4], Mat4 *Kinv);

struct ZnSpec {      // compact cache + the full-res geometry it encodes
    const float *zn = nullptr; int H = 0, W = 0; const float *K = nullptr;
    const int *frame_slot = nullptr;                       // persistent cache: device int[N], frame -> pool slot (B == 1 only)
    const uint32_t *lists = nullptr; const int *counts = nullptr;   // valid-pixel lists already built per slot
    const float *block_ranges = nullptr;                   // per (slot, 8 x 8 block) depth ranges already built (btba_zn_block_ranges / the pool)
};

static int solve_enqueue(btba_workspace *ws, const btba_params *prm, int B, int N, int Hd, int Wd, const float *intr,
                         const float *campos, const float *normals, const ZnSpec &Z, const btba_entryj *corr, int64_t corr_stride,
                         const uint32_t *pair_offsets, uint32_t max_corr_per_pair,
                         const int32_t *dense_pairs, int Pd_in, float *poses, float *trace, int *order_flag = nullptr,
                         bool corr24 = false, const uint32_t *pair_lens = nullptr)
{
    if (!ws || !prm || B < 1 || N < 2 || Hd < 2 || Wd < 2 || !intr || !poses) return BTBA_EINVAL;
    if (prm->n_gn_iters < 1 || prm->n_pcg_iters < 0) return BTBA_EINVAL;   // MLIB_ASSERT, CUDASolverBundling.cpp:194
    // A watchdog of an EARLIER chained launch fired and nobody has synchronised through the library since (a caller that orders its own stream with
    // btba_workspace_signal_stream never passes btba_workspace_sync): report it now -- that solve's poses were poisoned with NaN by the kernel (k_chain),
    // this and every later solve of the workspace runs unchained.
    // The device-visible word stays SET here: this path has not synchronised the stream, and the poison kernel queued behind the failed launch may not
    // have run yet -- clearing the word now would let it read 0 and leave that solve's garbage poses finite.  The word is cleared only behind a host
    // synchronisation (chain_check); `chain_reported` latches that the caller has been told, chaining is off from here on (only chained launches queue
    // poison kernels, so a sticky word cannot poison a later solve).
    if (ws->chain_error && *ws->chain_error && !ws->chain_reported) { ws->chain_reported = true; ws->chain_failed = true; return BTBA_ESCHED; }
    if (prm->reduction_mode != BTBA_REDUCE_DETERMINISTIC && prm->reduction_mode != BTBA_REDUCE_ATOMIC) return BTBA_EINVAL;
    const bool atomic_sums = prm->reduction_mode == BTBA_REDUCE_ATOMIC;      // the reference's way of summing (float atomics, order not fixed)
    if (atomic_sums && trace) return BTBA_EINVAL;                            // the decision traces are defined on the reproducible sums
    if (N > BTBA_MAX_FRAMES) return BTBA_EINVAL;                            // MAX_NUM_IMAGES of the reference (GlobalDefines.h:8) is 85 as well
    const int P = N * (N - 1) / 2;
    // Weights of the two terms per Gauss-Newton iteration: the solveBundlingStub seam takes one array per term (input.weightsSparse[nIter],
    // weightsDenseDepth[nIter], SolverBundling.cu:948-951; SBA.cpp:27-32 fills them with 1 / 1), and so do the optional arrays of btba_params.
    const float *wsi = prm->weights_sparse_per_iter, *wdi = prm->weights_dense_per_iter;
    if ((wsi || wdi) && prm->n_weights_per_iter != prm->n_gn_iters) return BTBA_EINVAL;      // the arrays' stated length: never read past it
    auto ws_at = [&](int it) { return wsi ? wsi[it] : prm->weight_sparse; };
    auto wd_at = [&](int it) { return wdi ? wdi[it] : prm->weight_dense_depth; };
    bool any_dense_weight = false, any_sparse_weight = false;
    for (int it = 0; it < prm->n_gn_iters; it++) {
        if (!(ws_at(it) >= 0.0f) || !(wd_at(it) >= 0.0f)) return BTBA_EINVAL;
        any_dense_weight |= wd_at(it) > 0.0f; any_sparse_weight |= ws_at(it) > 0.0f;
    }
    // The sparse sweep runs whenever a sparse weight is positive in SOME iteration; in an iteration whose weight is 0 it still runs, as in the
    // reference: evalMinusJTFDevice walks the correspondences for the Jacobi preconditioner whatever weightSparse is (SolverBundlingEquationsLie.h:
    // 60-137 -- the diagonal carries no weight factor), while right-hand side and operator take the factor 0.
    const bool use_sparse = any_sparse_weight && corr && pair_offsets && max_corr_per_pair > 0;
    // dense pair list
    std::vector<int32_t> pairs;
    if (any_dense_weight) {
        if (dense_pairs && Pd_in >= 0) pairs.assign(dense_pairs, dense_pairs + 2 * (size_t)Pd_in);
        else for (int i = 0; i < N; i++) for (int j = i + 1; j < N; j++) { pairs.push_back(i); pairs.push_back(j); }
        for (size_t k = 0; k < pairs.size(); k += 2)
            if (pairs[k] < 0 || pairs[k] >= N || pairs[k + 1] < 0 || pairs[k + 1] >= N || pairs[k] == pairs[k + 1]) return BTBA_EINVAL;
        if (dense_pairs && Pd_in >= 0) {
            // an explicit list may name a pair in both directions, but not the same (target, source) twice: the system's diagonal blocks would
            // count it twice and the one-pass off-diagonal assembly once (the reference's own list never repeats a pair, SolverBundling.cu:17-47)
            std::vector<char> seen((size_t)N * N, 0);
            for (size_t k = 0; k < pairs.size(); k += 2) {
                char &c = seen[(size_t)pairs[k] * N + pairs[k + 1]];
                if (c) return BTBA_EINVAL;
                c = 1;
            }
        }
    }
    const int Pd = (int)(pairs.size() / 2);
    const bool use_zn = Z.zn != nullptr;
    const bool use_dense = Pd > 0 && ((campos && normals) || use_zn);      // Pd == 0: "no overlapping images", SolverBundling.cu:280-283
    if (!use_sparse && !use_dense) {
        // nothing to optimise: poses still go through Log/Exp like the reference (SBA.cpp:106,115)
    }
    const int npix = Hd * Wd;
    // the launches are per instance GROUP when BTBA_FLAG_OVERLAP splits the batch over streams (below): the fuse test (>= 64 items of each kind) and the
    // chip-filling rules see a group's instances, so the partial counts are chosen for the smallest group
    int B_launch = B;
    if (B >= 8 && (prm->flags & BTBA_FLAG_OVERLAP)) {
        const int groups = std::max(1, std::min({ ws->tune.overlap_groups, (int)btba_workspace::kMaxGroups, B / 2 }));
        B_launch = B / groups;
    }
    const int chunks = use_sparse ? pick_chunks(prm, B_launch, P, max_corr_per_pair, use_dense && !(prm->flags & BTBA_FLAG_NO_FUSE)) : 1;
    const int tiles = use_dense ? pick_tiles(prm, B_launch, Pd, npix, use_zn && (prm->flags & BTBA_FLAG_COMPACTION), Wd, Hd) : 1;
    const bool timing = (prm->flags & BTBA_FLAG_TIME_KERNELS) != 0;
    const int timed_iteration = (prm->flags & BTBA_FLAG_TIME_SAMPLED) ? (int)(ws->solves_enqueued++ % (uint64_t)std::max(1, prm->n_gn_iters)) : -1;      // -1: all

    int rc;
    if ((rc = ws->x.ensure(sizeof(float) * 6 * (size_t)B * N))) return rc;
    if ((rc = ws->T.ensure(sizeof(float) * 16 * (size_t)B * N))) return rc;
    if ((rc = ws->Tinv.ensure(sizeof(float) * 16 * (size_t)B * N))) return rc;
    if ((rc = ws->sparse_part.ensure(sizeof(float) * (size_t)B * P * chunks * kSparseVals))) return rc;
    if ((rc = ws->dense_part.ensure(sizeof(float) * (size_t)B * (Pd > 0 ? Pd : 1) * tiles * kDenseVals))) return rc;
    // device table: [Pd x (target, source)] [N+1 adjacency offsets] [2 Pd adjacency entries = pair << 1 | is_source],
    // adjacency of a frame listed in pair order (fixed summation order)
    if (Pd > 0 && (pairs != ws->dense_pairs_host || ws->dense_pairs_frames != N)) {
        std::vector<int32_t> tab(pairs);
        std::vector<int32_t> off(N + 1, 0);
        for (int q = 0; q < Pd; q++) { off[pairs[2 * q] + 1]++; off[pairs[2 * q + 1] + 1]++; }
        for (int k = 0; k < N; k++) off[k + 1] += off[k];
        std::vector<int32_t> adj(2 * (size_t)Pd), cur(off.begin(), off.end() - 1);
        for (int q = 0; q < Pd; q++) { adj[cur[pairs[2 * q]]++] = (q << 1); adj[cur[pairs[2 * q + 1]]++] = (q << 1) | 1; }
        tab.insert(tab.end(), off.begin(), off.end());
        tab.insert(tab.end(), adj.begin(), adj.end());
        // canonical pair (i < j) -> the dense pair listed as (target i, source j), the one whose cross block the system keeps (-1: none)
        std::vector<int32_t> first_listed((size_t)N * N, -1);
        for (int q = Pd - 1; q >= 0; q--) first_listed[(size_t)pairs[2 * q] * N + pairs[2 * q + 1]] = q;
        for (int i = 0; i < N; i++) for (int j = i + 1; j < N; j++) tab.push_back(first_listed[(size_t)i * N + j]);
        // work order of the fused sweep's dense items: frames close in the window overlap most (keyframes are in temporal order), so
        // pairs sorted by |i - j| put the long workgroups first and the short ones at the end of the launch, where they drain quickly
        std::vector<int32_t> order(Pd);
        for (int q = 0; q < Pd; q++) order[q] = q;
        if (ws->tune.dense_order) std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b2) { return std::abs(pairs[2 * a] - pairs[2 * a + 1]) < std::abs(pairs[2 * b2] - pairs[2 * b2 + 1]); });
        // is that order the closed form the pinhole sweeps can compute for themselves?  (all pairs by ascending |i - j|, then ascending
        // lower frame, every pair's index = its canonical index, target = the lower (1) or the higher (2) frame throughout)
        ws->work_formula = 0;
        if (Pd == N * (N - 1) / 2) {
            for (int form = 1; form <= 2 && !ws->work_formula; form++) {
                bool same = true;
                int q = 0;
                for (int d = 1; d < N && same; d++)
                    for (int i = 0; i + d < N && same; i++, q++) {
                        const int pc = pair_index(N, (uint32_t)i, (uint32_t)(i + d));
                        same = order[q] == pc && pairs[2 * pc] == (form == 1 ? i : i + d) && pairs[2 * pc + 1] == (form == 1 ? i + d : i);
                    }
                if (same) ws->work_formula = form;
            }
        }
        while (tab.size() % 4) tab.push_back(0);           // 16-byte entries
        ws->dense_work_offset = tab.size();
        for (int q = 0; q < Pd; q++) { tab.push_back(pairs[2 * order[q]]); tab.push_back(pairs[2 * order[q] + 1]); tab.push_back(order[q]); tab.push_back(0); }
        if ((rc = ws->dense_pairs.ensure(sizeof(int32_t) * tab.size()))) return rc;
        HIP_TRY(hipMemcpyAsync(ws->dense_pairs.p, tab.data(), sizeof(int32_t) * tab.size(), hipMemcpyHostToDevice, ws->stream));
        HIP_TRY(hipStreamSynchronize(ws->stream));         // `tab` is a local; the copy must land before it dies
        ws->dense_pairs_host = pairs;
        ws->dense_pairs_frames = N;
    }
    // tables of k_system_solve that depend on the window size only: canonical pair p -> (i << 8 | j), then the 72
    // descriptors of the sparse 6x6 block entries (btba_kernels.hpp: sparse_entry_descriptor)
    if (ws->solve_tab_frames != N) {
        const size_t lut16 = ((size_t)P + 288 + 3) & ~(size_t)3;           // a second, 16-byte aligned copy of the descriptors (k_solve_small loads them as int4)
        std::vector<int32_t> tab(lut16 + 288);
        int q = 0;
        for (int i = 0; i < N; i++) for (int j = i + 1; j < N; j++) tab[q++] = (i << 8) | j;
        for (int t = 0; t < 72; t++) { sparse_entry_descriptor(t >= 36, (t % 36) / 6, t % 6, &tab[(size_t)P + 4 * t]); sparse_entry_descriptor(t >= 36, (t % 36) / 6, t % 6, &tab[lut16 + 4 * t]); }
        if ((rc = ws->solve_tab.ensure(sizeof(int32_t) * tab.size()))) return rc;
        HIP_TRY(hipMemcpyAsync(ws->solve_tab.p, tab.data(), sizeof(int32_t) * tab.size(), hipMemcpyHostToDevice, ws->stream));
        HIP_TRY(hipStreamSynchronize(ws->stream));         // `tab` is a local
        ws->solve_tab_frames = N;
    }
    const int32_t *d_adj_off = ws->dense_pairs.as<int32_t>() + 2 * (size_t)Pd;
    const int32_t *d_adj = d_adj_off + (N + 1);

    SolveDims D{};
    D.n_frames = N; D.n_pairs = P; D.n_dense_pairs = use_dense ? Pd : 0;
    D.npix = npix; D.width = Wd; D.height = Hd;
    D.sparse_chunks = chunks; D.dense_tiles = tiles; D.n_pcg = prm->n_pcg_iters;
    D.use_sparse = use_sparse; D.use_dense = use_dense;
    D.fx = intr[0]; D.fy = intr[1]; D.cx = intr[2]; D.cy = intr[3];
    D.robust_delta = prm->robust_delta; D.dist_thresh = prm->dense_dist_thresh; D.normal_thresh = prm->dense_normal_thresh;
    D.dist2_thresh = prm->dense_dist_thresh * prm->dense_dist_thresh;
    D.wm1 = (float)(Wd - 1); D.hm1 = (float)(Hd - 1); D.wm2 = (float)(Wd - 2); D.hm2 = (float)(Hd - 2);
    D.depth_min = prm->depth_min; D.depth_max = prm->depth_max;
    D.w_sparse = prm->weight_sparse; D.w_dense = prm->weight_dense_depth;
    D.corr_stride = corr_stride;
    D.corr_nt_from = B;                // plain loads unless decided otherwise below
    D.order_flag = order_flag;
    D.live_blocks = (ws->count_live && ws->live_blocks.p) ? ws->live_blocks.as<unsigned long long>() : nullptr;
    D.pose_stride = 16 * N; D.x_stride = 6 * N;                      // instances back to back (the chained launch pads them, below)
    D.sp_stride = (unsigned)((size_t)P * (atomic_sums ? 1 : chunks) * kSparseVals);
    D.dp_stride = (unsigned)((size_t)(Pd > 0 ? Pd : 1) * (atomic_sums ? 1 : tiles) * kDenseVals);
    if ((size_t)B * std::max(D.sp_stride, D.dp_stride) >= ((size_t)1 << 32) || (size_t)B * 16 * N >= ((size_t)1 << 31)) return BTBA_EINVAL;      // the sweeps address an instance's poses and partials in 32 bits
    D.atomic_sums = atomic_sums ? 1 : 0;
    D.corr24 = corr24 ? 1 : 0;
    D.pair_lens = pair_lens;
    if (Pd > 0) { D.dense_work = reinterpret_cast<const int4 *>(ws->dense_pairs.as<int32_t>() + ws->dense_work_offset); D.work_formula = ws->work_formula; }      // work position -> (target, source, pair, -)
    D.tile_major = ws->tune.tile_major ? 1 : 0;
    D.sparse_tail_256 = 0;             // set below, once the dense layout is known
    D.walk_blocks = (Wd % 8 == 0 && Hd % 8 == 0 && Wd + Hd <= 1024 && ws->tune.block_walk) ? 1 : 0;
    D.n_gn = prm->n_gn_iters;
    int zn_layout = 0;
    if (use_zn) {
        if (!Z.K || Z.H < 2 || Z.W < 2) return BTBA_EINVAL;
        float intr_chk[4];
        Mat4 Kinv;
        scaled_intrinsics(Z.H, Z.W, Hd, Wd, Z.K, intr_chk, &Kinv);
        for (int k = 0; k < 16; k++) D.zn_ki[k] = Kinv.m[k];
        D.zn_scale_w = (float)(Z.W - 1) / (float)(Wd - 1);
        D.zn_scale_h = (float)(Z.H - 1) / (float)(Hd - 1);
        const float *ki = D.zn_ki;
        D.zn_simple = (ki[1] == 0.f && ki[3] == 0.f && ki[4] == 0.f && ki[7] == 0.f && ki[12] == 0.f && ki[13] == 0.f && ki[14] == 0.f && ki[15] == 1.0f) ? 1 : 0;     // pinhole: z = d exactly
        zn_layout = D.zn_simple ? 1 : 2;
    }
    D.trace_on = (trace != nullptr) && (prm->flags & BTBA_FLAG_TRACE);
    btba_trace_layout L;
    btba_trace_layout_get(N, D.n_dense_pairs, prm->n_pcg_iters, &L);
    D.trace_record = L.record_floats; D.tr_x = L.off_x; D.tr_T = L.off_T; D.tr_rhs = L.off_rhs; D.tr_prec = L.off_precond;
    D.tr_pcg = L.off_pcg; D.tr_delta = L.off_delta; D.tr_dpair = L.off_dense_pair; D.tr_A = L.off_A; D.tr_clk = L.off_clk;

    const size_t n = 6 * (size_t)N, ld = 4 * (((n + 3) / 4) | 1);
    // the 6N x 6N matrix lives in the CU's LDS when it fits (N <= BTBA_MAX_FRAMES_LDS with the default pair list); larger
    // windows (up to the reference's 85 frames) keep it in an L2-resident global scratch and run the multi-wave PCG
    const size_t lds_rest = (6 * ld + 16 + 16 * (size_t)N + 4 * (size_t)D.n_dense_pairs + (size_t)N + 1 + (size_t)P + 288 + (((size_t)P + 3) & ~(size_t)3) + ((6 * (size_t)N + 3) & ~(size_t)3)) * sizeof(float);
    const size_t lds_limit = 160 * 1024;
    const bool a_global = n * ld * sizeof(float) + lds_rest > lds_limit;
    const size_t lds_core = (a_global ? 0 : n * ld * sizeof(float)) + lds_rest;
    const size_t lds_pairs = ((size_t)P * kSparseVals + (size_t)D.n_dense_pairs * kDenseVals) * sizeof(float);
    D.pairsum_in_lds = (lds_core + lds_pairs <= 64 * 1024) ? 1 : 0;   // keep two workgroups per CU when it fits
    if (!D.pairsum_in_lds && lds_core + lds_pairs <= lds_limit && B <= 256) D.pairsum_in_lds = 1;
    const size_t lds_bytes = lds_core + (D.pairsum_in_lds ? lds_pairs : 0);
    if (lds_bytes > lds_limit) return BTBA_EINVAL;
    if (!D.pairsum_in_lds) { if ((rc = ws->pairsum.ensure((size_t)B * lds_pairs + 16))) return rc; }
    // larger windows: reduce the partials and assemble the system on many workgroups (k_big_reduce, k_big_assemble); the traced solve
    // keeps the single-workgroup path, whose trace records the system.  Worth two extra launches per iteration from ~24 frames on
    // (one workgroup: 90 k of the 140 k cycles of a launch at N = 31 are reduction and assembly; at N = 15 19 k of 45 k, less than the launches)
    // (round 6: windows of 22 ... 31 frames with at most one dense pair per canonical pair run k_solve_mid -- reduction, assembly, PCG and update in ONE launch,
    // btba_solve_mid.hpp -- unless BTBA_OPT_SOLVE_SMALL = 0 or the reference's atomic summation is asked for)
    const size_t mid_lds = sizeof(float) * mid_solve_lds_floats(N, D.n_dense_pairs);
    const bool mid_solve = ws->tune.solve_small && N > kSmallMaxFrames && N <= kMidMaxFrames && !atomic_sums && !a_global && D.n_dense_pairs <= P && mid_lds <= lds_limit;
    D.pre_assembled = ((a_global || N >= 24) && !trace && ws->tune.big_assembly && !mid_solve) ? 1 : 0;
    if (a_global || D.pre_assembled) { if ((rc = ws->big_A.ensure((size_t)B * (n + 2) * ld * sizeof(float)))) return rc; }      // per instance: A[n][ld], rhs[ld], prec[ld]
    if (D.pre_assembled && D.pairsum_in_lds) { D.pairsum_in_lds = 0; if ((rc = ws->pairsum.ensure((size_t)B * lds_pairs + 16))) return rc; }
    if (lds_bytes > 64 * 1024 || !ws->lds_attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_system_solve<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_system_solve<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_system_solve<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_system_solve<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        ws->lds_attr_set = true;
    }

    // tracker-sized windows run k_solve_small (btba_solve_small.hpp) instead of k_system_solve: not in atomic mode (its records are accumulators
    // the solve has to clear), not for the chained launch's last solve, and only while everything fits the CU's LDS
    const int small_cpl_v = small_cpl(N);
    const size_t small_lds = sizeof(float) * small_solve_lds_floats(N, D.n_dense_pairs, small_cpl_v);
    const bool small_solve = ws->tune.solve_small && N <= kSmallMaxFrames && !atomic_sums && !a_global && !D.pre_assembled && 2 * D.n_dense_pairs <= kSmallBlock && small_lds <= lds_limit;
    if (small_solve && !ws->small_attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_solve_small<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_solve_small<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_solve_small<12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_solve_small<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        ws->small_attr_set = true;
    }
    if (mid_solve && !ws->mid_attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_solve_mid), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        ws->mid_attr_set = true;
    }

    // stats bookkeeping (collected after sync)
    btba_stats &S = ws->stats;
    if (ws->events.empty() || !timing) std::memset(&S, 0, sizeof S);     // new accumulation window
    S.n_instances = B; S.n_frames = N; S.n_pairs = P; S.n_dense_pairs = D.n_dense_pairs;
    S.dense_tiles = tiles; S.sparse_chunks = chunks;
    S.bytes_dense_alg = use_dense ? (int64_t)64 * D.n_dense_pairs * npix * B : 0;
    S.bytes_sparse_alg = 0;   // filled by callers that know C (optimize_frames) or from offsets on request

    {   // whether the two sweeps ride in one launch (same rule as in the loop below), reported whatever the timing flags
        const int nb0 = (B >= 8 && (prm->flags & BTBA_FLAG_OVERLAP)) ? B / 2 : B;
        S.fused_sweeps = (use_sparse && use_dense && (unsigned)chunks * P * nb0 >= 64u && (unsigned)tiles * D.n_dense_pairs * nb0 >= 64u && !(prm->flags & BTBA_FLAG_NO_FUSE)) ? 1 : 0;
    }
    size_t reg;
    if ((rc = time_begin(ws, timing || ws->always_time_region, 3, &reg))) return rc;
    if (atomic_sums) {        // one accumulator record per pair: empty before the first sweep (k_system_solve leaves them empty again)
        HIP_TRY(hipMemsetAsync(ws->sparse_part.p, 0, sizeof(float) * (size_t)B * P * kSparseVals, ws->stream));
        HIP_TRY(hipMemsetAsync(ws->dense_part.p, 0, sizeof(float) * (size_t)B * (Pd > 0 ? Pd : 1) * kDenseVals, ws->stream));
    }
    const bool compaction = use_zn && use_dense && (prm->flags & BTBA_FLAG_COMPACTION);
    // The chained launch (btba_kernels.hpp: k_chain): all Gauss-Newton iterations in ONE launch, the system solves handed over inside it.
    // Same sums in the same order as the plain schedule; taken for the batches it pays for and the configurations it is written for.
    const int chain_lay = zn_layout == 1 ? (compaction ? 3 : 1) : 0;
    // Only on request (BTBA_OPT_CHAIN = 1).  Measured at c3 x 32 on one box (profiles/r04/chain_experiments.json): the chained launch with two tiles
    // per pair 1.280 ms per step against 1.306 ms for the plain schedule with two tiles -- and 1.247 ms for the plain schedule with ONE tile, which the
    // chained launch cannot use (1.314 ms: its sparse items come in bursts behind every instance's dense items instead of closing the launch).
    // Object-masked frames: 0.74 against 0.52 ms (their sweeps are shorter than an in-launch solve).  The library's own choice is the plain schedule.
    // (only on a device that shows all 8 XCDs x 32 compute units -- the launch's forward progress rests on workgroup g running on XCD g % 8 in grid order;
    // a partitioned device runs the plain schedule)
    const bool chain = chain_lay != 0 && !ws->chain_failed && ws->n_cus == 256 && ws->tune.chain > 0 && use_sparse && use_dense && !wsi && !wdi
                       && !trace && !atomic_sums && !a_global && !D.pre_assembled && N <= kChainMaxFrames && chunks <= kChainMaxParts && tiles <= kChainMaxParts
                       && !(prm->flags & (BTBA_FLAG_NO_FUSE | BTBA_FLAG_OVERLAP)) && !Z.frame_slot
                       && lds_rest + 16 + sizeof(float) * chain_region_floats(N) <= kChainLdsBytes;      // (c3's 15 frames are the largest window whose solve fits a sweep workgroup's LDS share)
    // Fresh matches arrive as EntryJ (32 B, the wire format).  A solve reads them once per Gauss-Newton iteration; with BTBA_OPT_RELAYOUT the FIRST
    // iteration's sparse sweep writes 24-byte records as the entries pass by (SolveDims::corr24_out) and the other iterations stream those -- no separate
    // pack pass (btba_pack_correspondences24: 65 us + a launch at c3 x 32, 7 % of a step).  Measured (profiles/r04/relayout.json): the first launch's
    // extra 161 MB of writes (+55 us, the launch's sparse tail is HBM-bound) cost what six launches save by reading 24 instead of 32 bytes:
    // 1.286 / 1.300 ms per step against 1.297 / 1.287 ms with EntryJ in every iteration on two boxes -- a wash, so it is OFF by default.
    // (not on object-masked frames walked through valid-pixel lists: their fused launch is short and latency-bound, and it measured FASTER on the
    // 32-byte entries -- 49.1 against 51.2 us, profiles/r03 -- before the re-layout's extra write in the first iteration: 454 k -> 402 k GN it/s with it)
    const bool relayout = ws->tune.relayout && use_sparse && !corr24 && !pair_lens && !chain && prm->n_gn_iters >= 3 && !atomic_sums && !compaction
                          && (size_t)B * (size_t)corr_stride * sizeof(btba_entryj) >= ((size_t)4 << 20);
    if (relayout) { if ((rc = ws->corr24_tmp.ensure(sizeof(float2) * 192 * (((size_t)B * (size_t)corr_stride + 63) / 64)))) return rc; }
    // Non-temporal correspondence loads (SolveDims::corr_nt).  Every iteration re-reads the batch's frames (dense items, through L2) and streams its
    // correspondences once.  While both fit the memory-side cache (256 MB) the next iteration finds them there and plain loads are fastest (c3 x 8,
    // c2 x 32, object-masked frames, which touch a few per cent of their pixels: non-temporal loads measured -2 ... -17 %); beyond it the cyclic stream
    // evicts the frames, and reading it non-temporally keeps them resident: c3 x 32 (147 + 161 MB) 185.2 -> 195.2 k GN it/s, c4 x 32 +3.7 %
    // (profiles/r04/nontemporal_stream.json).  Same bits.
    const long long frames_bytes = use_dense ? (long long)B * N * npix * (use_zn ? 16 : 32) : 0;
    const long long corr_bytes = use_sparse ? (long long)B * (long long)corr_stride * (corr24 ? 24 : 32) : 0;
    const long long corr_per_instance = corr_bytes / B;
    // instances from this index on stream non-temporally: all that does not fit beside the frames (everything when the frames alone exceed the cache)
    int corr_nt_auto = B;
    if (use_dense && use_sparse && !compaction && frames_bytes + corr_bytes > ws->tune.last_level_cache)        // (object-masked frames: measured slower at c3 x 32 whatever the layout)
        corr_nt_auto = (frames_bytes >= ws->tune.last_level_cache || corr_per_instance <= 0) ? 0 : (int)std::min<long long>(B, (ws->tune.last_level_cache - frames_bytes) / corr_per_instance);
    if (ws->tune.corr_nt_partial == 0 && corr_nt_auto < B) corr_nt_auto = 0;
    const int corr_nt_from = ws->tune.corr_nt >= 0 ? (ws->tune.corr_nt ? 0 : B) : corr_nt_auto;
    ChainDims Cn{};
    int plain_pairsum_in_lds = 0;
    auto pad32 = [](size_t floats) { return (floats + 31) & ~(size_t)31; };      // whole 128-byte lines
    if (chain) {
        const int n_it = prm->n_gn_iters;
        D.pose_stride = (int)pad32(16 * (size_t)N); D.x_stride = (int)pad32(6 * (size_t)N);
        D.sp_stride = (unsigned)pad32((size_t)P * chunks * kSparseVals); D.dp_stride = (unsigned)pad32((size_t)Pd * tiles * kDenseVals);
        D.publish = 1;
        D.corr_nt_from = corr_nt_from;
        plain_pairsum_in_lds = D.pairsum_in_lds;
        D.pairsum_in_lds = 0;
        Cn.last_solve_external = 1;
        Cn.n_iter = n_it; Cn.n_inst = B; Cn.inst_per_xcd = (unsigned)((B + 7) / 8);
        Cn.items_d = (unsigned)tiles * (unsigned)D.n_dense_pairs; Cn.items_s = (unsigned)chunks * (unsigned)P;
        Cn.sparse_period = (unsigned)ws->tune.chain_sparse_period;
        if (Cn.sparse_period >= 2u && Cn.sparse_period * Cn.items_s > Cn.items_d + Cn.items_s) Cn.sparse_period = (Cn.items_d + Cn.items_s) / Cn.items_s;      // all sparse items must find a slot
        if (Cn.sparse_period < 2u) Cn.sparse_period = 0u;
        Cn.group = (Cn.sparse_period == 0u && ws->tune.chain_group > 1 && Cn.inst_per_xcd % (unsigned)ws->tune.chain_group == 0u && B % 8 == 0) ? (unsigned)ws->tune.chain_group : 1u;
        Cn.timeout_ticks = (long long)ws->tune.chain_timeout_ms * 100000ll;
        Cn.solve_prio = ws->tune.chain_solve_prio;
        Cn.debug_skip = ws->tune.chain_debug_skip;
        if (Cn.debug_skip & 16) D.publish = 0;             // (timing experiment: plain record stores)
        Cn.pose_ring = (size_t)B * D.pose_stride; Cn.x_ring = (size_t)B * D.x_stride;
        Cn.sp_ring = (size_t)B * (size_t)D.sp_stride; Cn.dp_ring = (size_t)B * (size_t)D.dp_stride;
        Cn.ps_size = pad32((size_t)D.n_dense_pairs * kDenseVals); Cn.A_size = pad32((n + 2) * ld);      // (global scratch: the reduced DENSE pair sums; the matrix before it is packed into LDS)
        if ((rc = ws->x.ensure(sizeof(float) * (n_it + 1) * Cn.x_ring))) return rc;
        if ((rc = ws->T.ensure(sizeof(float) * (n_it + 1) * Cn.pose_ring))) return rc;
        if ((rc = ws->Tinv.ensure(sizeof(float) * (n_it + 1) * Cn.pose_ring))) return rc;
        if ((rc = ws->sparse_part.ensure(sizeof(float) * n_it * Cn.sp_ring))) return rc;
        if ((rc = ws->dense_part.ensure(sizeof(float) * n_it * Cn.dp_ring))) return rc;
        if ((rc = ws->pairsum.ensure(sizeof(float) * (size_t)n_it * B * Cn.ps_size))) return rc;
        if ((rc = ws->big_A.ensure(sizeof(float) * (size_t)n_it * B * Cn.A_size))) return rc;
        const size_t sync_head = ((size_t)B * (n_it + 1) + 15) & ~(size_t)15;               // flags[B], arrivals[n_it][B], padded to a 64-byte line
        const size_t sync_ints = sync_head + (size_t)16 * B * (n_it + 1);                  // ... published[n_it + 1][B][16]
        if ((rc = ws->chain_sync.ensure(sizeof(int) * sync_ints))) return rc;
        if (!ws->chain_error) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ws->chain_error), 64, hipHostMallocDefault));
            *ws->chain_error = 0;
        }
        HIP_TRY(hipMemsetAsync(ws->chain_sync.p, 0, sizeof(int) * sync_ints, ws->stream));      // flags and arrival counters: zero before EVERY launch
        Cn.flags = ws->chain_sync.as<int>(); Cn.arrivals = Cn.flags + B; Cn.published = Cn.flags + sync_head; Cn.error = ws->chain_error;
    }
    // Log, Exp, inverse of the incoming matrices
    {
        const int total = B * N;
        if (chain) k_prepare_strided<<<(total + 63) / 64, 64, 0, ws->stream>>>(total, N, D.pose_stride, D.x_stride, poses, ws->x.as<float>(), ws->T.as<float>(), ws->Tinv.as<float>());
        else k_prepare<<<(total + 63) / 64, 64, 0, ws->stream>>>(total, poses, ws->x.as<float>(), ws->T.as<float>(), ws->Tinv.as<float>(), ws->tune.prepare_keep_T);
    }
    // per-frame valid-pixel lists for the compact dense sweep (once per solve; the frames do not change across iterations)
    const uint32_t *lists_base = Z.lists;
    const int *counts_base = Z.counts;
    if (compaction && !lists_base) {
        if ((rc = ws->valid_lists.ensure(sizeof(uint32_t) * (size_t)B * N * npix))) return rc;
        if ((rc = ws->valid_counts.ensure(sizeof(int) * (size_t)B * N))) return rc;
        k_valid_lists<<<B * N, 1024, 0, ws->stream>>>(npix, reinterpret_cast<const float4 *>(Z.zn), ws->valid_lists.as<uint32_t>(), ws->valid_counts.as<int>(), nullptr);
        lists_base = ws->valid_lists.as<uint32_t>();
        counts_base = ws->valid_counts.as<int>();
    }
    if (Z.frame_slot && B != 1) return BTBA_EINVAL;
    D.frame_slot = Z.frame_slot;
    // block walk of the pinhole sweep: per-block depth ranges for its dead-block test, and the LDS it needs for the list of live blocks
    size_t blist_bytes = 0;
    if (!(zn_layout == 1 && use_dense && !compaction)) D.walk_blocks = 0;
    // where the sparse items go in the fused launch: all of them at the END when the dense items are the long, VALU-bound walks over full
    // frames (c3 x 32: 176.4 -> 170.5 us per launch, gpurun_out/r03_24: they fill the launch's drain); interleaved when the dense items are
    // the short list walks of object-masked frames (49.7 interleaved against 53.3 us)
    D.sparse_tail_256 = ws->tune.sparse_tail_256 >= 0 ? ws->tune.sparse_tail_256 : ((use_dense && !compaction) ? 256 : 0);
    if (D.walk_blocks) {
        const int bw = Wd / 8, bh = Hd / 8;
        const size_t band_blocks = (size_t)((bh + tiles - 1) / tiles) * bw;
        if (band_blocks > 1024) D.walk_blocks = 0;                               // very large caches: row strips (a band's blocks are tested by 4 x 256 lanes)
        else {
            blist_bytes = 32 + sizeof(uint32_t) * band_blocks;
            if (ws->tune.block_skip) {                                          // (off: every block is walked)
                if (Z.block_ranges) D.block_ranges = reinterpret_cast<const float2 *>(Z.block_ranges);      // part of the caller's / the pool's frame cache
                else if (!Z.frame_slot) {                                       // not given: one pass over the frames per solve
                    if ((rc = ws->block_ranges.ensure(sizeof(float2) * (size_t)B * N * bw * bh))) return rc;
                    k_block_ranges<<<dim3((unsigned)((bw * bh + kBlock / 64 - 1) / (kBlock / 64)), (unsigned)(B * N)), kBlock, 0, ws->stream>>>(Wd, Hd, reinterpret_cast<const float4 *>(Z.zn), nullptr, ws->block_ranges.as<float2>());
                    D.block_ranges = ws->block_ranges.as<float2>();
                }
            }
        }
    }
    // Software pipelining across instances: the batch is split in two halves on two streams, so one half's
    // latency-bound k_system_solve (B/2 workgroups on a 256-CU chip) and its sparse sweep overlap the other
    // half's dense sweep.  Halves never touch each other's data; fork/join events keep the caller's stream
    // ordering.  Small batches run as one piece.
    int n_halves = (B >= 8 && (prm->flags & BTBA_FLAG_OVERLAP)) ? 2 : 1;
    if (n_halves == 2) n_halves = std::max(1, std::min({ ws->tune.overlap_groups, (int)btba_workspace::kMaxGroups, B / 2 }));
    for (int g = 1; g < n_halves; g++) {
        if (ws->aux_streams[g - 1]) continue;
        // LOWEST priority by default: equal-priority streams with identical kernel sequences were measured to time-share the chip in
        // lockstep; with a priority gap the main group is never held up and the others fill the CUs it leaves idle.
        int prio_least = 0, prio_greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        HIP_TRY(hipStreamCreateWithPriority(&ws->aux_streams[g - 1], hipStreamNonBlocking, ws->tune.overlap_equal_prio ? 0 : prio_least));
        HIP_TRY(hipEventCreateWithFlags(&ws->ev_join[g - 1], hipEventDisableTiming));
        if (!ws->ev_fork) HIP_TRY(hipEventCreateWithFlags(&ws->ev_fork, hipEventDisableTiming));
    }
    struct Half { int b0, nb; hipStream_t st; } halves[btba_workspace::kMaxGroups];
    for (int g = 0; g < n_halves; g++) {
        const int lo = (int)((long long)B * g / n_halves), hi = (int)((long long)B * (g + 1) / n_halves);
        halves[g] = { lo, hi - lo, g == 0 ? ws->stream : ws->aux_streams[g - 1] };
    }
    if (n_halves > 1) {
        HIP_TRY(hipEventRecord(ws->ev_fork, ws->stream));
        for (int g = 1; g < n_halves; g++) HIP_TRY(hipStreamWaitEvent(ws->aux_streams[g - 1], ws->ev_fork, 0));
    }
    const size_t pairsum_floats = lds_pairs / sizeof(float);
    S.chain_iterations = 0;
    const size_t lut_bytes = sizeof(float) * (size_t)((Wd + Hd + 3) & ~3) + (zn_layout == 1 ? sizeof(float4) * (size_t)(Wd + Hd + 4) : 0) + blist_bytes + (size_t)ws->tune.debug_lds_pad;         // coordinate look-up tables of the compact dense sweep (+ its list of live blocks)
    if (chain) {
        // ONE launch: 8 XCD sequences x n_gn iterations x (instances of the XCD) x (sweep items + 1 solve item)
        const size_t chain_lds = std::max(lut_bytes, lds_rest + 16 + sizeof(float) * chain_region_floats(N));
        const unsigned per_inst = Cn.items_d + Cn.items_s + 1u;
        const unsigned grid = 8u * (unsigned)prm->n_gn_iters * Cn.inst_per_xcd * per_inst;
        if (!ws->tune.chain_trace_file.empty()) {          // developer: timeline of the launch (scripts/chain_trace.py)
            const size_t trace_bytes = 32 * (size_t)grid + 64 * (size_t)prm->n_gn_iters * B;
            if ((rc = ws->chain_trace.ensure(trace_bytes))) return rc;
            HIP_TRY(hipMemsetAsync(ws->chain_trace.p, 0, trace_bytes, ws->stream));
            Cn.trace = ws->chain_trace.as<unsigned long long>();
            Cn.stamps = Cn.trace + 4 * (size_t)grid;
        }
        const float4 *corr_c = reinterpret_cast<const float4 *>(corr);
        size_t slot;
        if (Cn.debug_skip & 32) {
            // developer TIMING experiment: the chain kernel as a plain per-iteration sweep launch (its solve items return at once) followed by
            // k_system_solve on ring slot 0 -- what the sweep items cost in THIS kernel binary without the in-launch schedule
            ChainDims C1 = Cn;
            C1.n_iter = 1; C1.debug_skip |= 8; C1.trace = nullptr; C1.stamps = nullptr;
            const unsigned grid1 = 8u * Cn.inst_per_xcd * per_inst;
            if ((rc = ws->pairsum.ensure(sizeof(float) * (size_t)B * ((size_t)P * kSparseVals + (size_t)D.n_dense_pairs * kDenseVals + 64)))) return rc;
            for (int it = 0; it < prm->n_gn_iters; it++) {
                if ((rc = time_begin(ws, timing, 0, &slot))) return rc;
#define BTBA_CHAIN_ARGS1 D, C1, reinterpret_cast<const float4 *>(Z.zn), ws->T.as<float>(), ws->Tinv.as<float>(), ws->x.as<float>(), ws->dense_part.as<float>(), corr_c, pair_offsets, \
                        ws->sparse_part.as<float>(), compaction ? lists_base : nullptr, compaction ? counts_base : nullptr, ws->dense_pairs.as<int2>(), d_adj_off, d_adj, ws->solve_tab.as<int>(), \
                        ws->pairsum.as<float>(), ws->big_A.as<float>(), poses
                if (chain_lay == 1) k_chain<1><<<dim3(grid1), kBlock, chain_lds, ws->stream>>>(BTBA_CHAIN_ARGS1);
                else k_chain<3><<<dim3(grid1), kBlock, chain_lds, ws->stream>>>(BTBA_CHAIN_ARGS1);
#undef BTBA_CHAIN_ARGS1
                if ((rc = time_end(ws, slot))) return rc;
                if ((rc = time_begin(ws, timing, 2, &slot))) return rc;
                k_system_solve<false, false><<<B, kSolveBlock, lds_core, ws->stream>>>(D, it, ws->sparse_part.as<float>(), ws->dense_part.as<float>(), ws->dense_pairs.as<int2>(), d_adj_off, d_adj,
                    ws->x.as<float>(), ws->T.as<float>(), ws->Tinv.as<float>(), ws->pairsum.as<float>(), nullptr, nullptr, it == prm->n_gn_iters - 1 ? poses : nullptr, ws->solve_tab.as<int>());
                if ((rc = time_end(ws, slot))) return rc;
            }
            S.fused_sweeps = 1;
        } else {
        if ((rc = time_begin(ws, timing, 0, &slot))) return rc;
#define BTBA_CHAIN_ARGS D, Cn, reinterpret_cast<const float4 *>(Z.zn), ws->T.as<float>(), ws->Tinv.as<float>(), ws->x.as<float>(), ws->dense_part.as<float>(), corr_c, pair_offsets, \
                        ws->sparse_part.as<float>(), compaction ? lists_base : nullptr, compaction ? counts_base : nullptr, ws->dense_pairs.as<int2>(), d_adj_off, d_adj, ws->solve_tab.as<int>(), \
                        ws->pairsum.as<float>(), ws->big_A.as<float>(), poses
        if (chain_lay == 1) k_chain<1><<<dim3(grid), kBlock, chain_lds, ws->stream>>>(BTBA_CHAIN_ARGS);
        else k_chain<3><<<dim3(grid), kBlock, chain_lds, ws->stream>>>(BTBA_CHAIN_ARGS);
#undef BTBA_CHAIN_ARGS
        if ((rc = time_end(ws, slot))) return rc;
        if (Cn.last_solve_external) {
            // the last iteration's system solve as its own launch on the last ring slot: the launch above has drained, 16 waves per instance
            const int il = prm->n_gn_iters - 1;
            SolveDims Dl = D;
            Dl.pairsum_in_lds = plain_pairsum_in_lds; Dl.publish = 0;
            if (!Dl.pairsum_in_lds) { if ((rc = ws->pairsum.ensure(std::max(sizeof(float) * (size_t)prm->n_gn_iters * B * Cn.ps_size, (size_t)B * lds_pairs + 16)))) return rc; }
            if ((rc = time_begin(ws, timing, 2, &slot))) return rc;
#define BTBA_LAST_SOLVE(LP) k_system_solve<LP, false><<<B, kSolveBlock, lds_bytes, ws->stream>>>(Dl, il, ws->sparse_part.as<float>() + il * Cn.sp_ring, ws->dense_part.as<float>() + il * Cn.dp_ring, \
                ws->dense_pairs.as<int2>(), d_adj_off, d_adj, ws->x.as<float>() + il * Cn.x_ring, ws->T.as<float>() + il * Cn.pose_ring, ws->Tinv.as<float>() + il * Cn.pose_ring, \
                ws->pairsum.as<float>(), nullptr, nullptr, poses, ws->solve_tab.as<int>())
            if (Dl.pairsum_in_lds) BTBA_LAST_SOLVE(true); else BTBA_LAST_SOLVE(false);
#undef BTBA_LAST_SOLVE
            if ((rc = time_end(ws, slot))) return rc;
        }
        // a watchdog that fired inside the launch leaves poses nobody may use: NaN them on the device, so that a caller who never synchronises through
        // the library (and so never sees BTBA_ESCHED from btba_workspace_sync) runs into its own finiteness checks / BTBA_ENUMERIC
        k_chain_poison<<<(16 * B * N + 255) / 256, 256, 0, ws->stream>>>(Cn.error, poses, 16 * B * N);
        S.chain_iterations = prm->n_gn_iters;
        S.fused_sweeps = 1;
        ws->chain_launches++;
        if (Cn.trace) {
            HIP_TRY(hipStreamSynchronize(ws->stream));
            std::vector<unsigned long long> host(4 * (size_t)grid + 8 * (size_t)prm->n_gn_iters * B);
            HIP_TRY(hipMemcpy(host.data(), ws->chain_trace.p, 8 * host.size(), hipMemcpyDeviceToHost));
            if (FILE *f = std::fopen(ws->tune.chain_trace_file.c_str(), "wb")) {       // [0] = number of block records, then the records, then the solve items' phase stamps
                const unsigned long long nrec = grid;
                std::fwrite(&nrec, 8, 1, f); std::fwrite(host.data(), 8, host.size(), f); std::fclose(f);
            }
        }
        }
    } else
    for (int it = 0; it < prm->n_gn_iters; it++) {
        const bool timing_it = timing && (timed_iteration < 0 || timed_iteration == it);
        // this iteration's weights (parameters.weightSparse / weightDenseDepth / useDense of SolverBundling.cu:948-953)
        SolveDims Di = D;
        Di.w_sparse = ws_at(it); Di.w_dense = wd_at(it);
        const bool use_dense_it = use_dense && Di.w_dense > 0.0f;
        Di.use_dense = use_dense_it ? 1 : 0;
        for (int h = 0; h < n_halves; h++) {
            const Half &H = halves[h];
            const size_t b0 = (size_t)H.b0;
            const float *campos_h = campos ? campos + 4 * b0 * N * npix : nullptr, *normals_h = normals ? normals + 4 * b0 * N * npix : nullptr;
            const float4 *zn_h = use_zn ? reinterpret_cast<const float4 *>(Z.zn) + b0 * N * npix : nullptr;
            const uint32_t *vl_h = compaction ? lists_base + b0 * N * npix : nullptr;
            const int *vc_h = compaction ? counts_base + b0 * N : nullptr;
            // (24-byte correspondences live in 64-entry groups counted from the start of the array: the half keeps the array's base and its first instance's entry offset)
            const bool c24_it = corr24 || (relayout && it > 0);      // this iteration reads 24-byte records (the caller's, or the ones iteration 0 wrote)
            const float4 *corr_h = corr ? (corr24 ? reinterpret_cast<const float4 *>(corr) : (relayout && it > 0) ? ws->corr24_tmp.as<float4>()
                                                  : reinterpret_cast<const float4 *>(corr) + 2 * b0 * (size_t)corr_stride) : nullptr;
            const uint32_t *off_h = pair_offsets ? pair_offsets + b0 * (P + 1) : nullptr;
            float *x_h = ws->x.as<float>() + 6 * b0 * N, *T_h = ws->T.as<float>() + 16 * b0 * N, *Ti_h = ws->Tinv.as<float>() + 16 * b0 * N;
            float *sp_h = ws->sparse_part.as<float>() + b0 * P * (atomic_sums ? 1 : chunks) * kSparseVals;
            float *dp_h = ws->dense_part.as<float>() + b0 * (size_t)(Pd > 0 ? Pd : 1) * (atomic_sums ? 1 : tiles) * kDenseVals;
            float *ps_h = ws->pairsum.p ? ws->pairsum.as<float>() + b0 * pairsum_floats : nullptr;
            float *tr_h = trace ? trace + b0 * (size_t)D.n_gn * D.trace_record : nullptr;
            size_t slot;
            SolveDims Dh = Di;                                  // the sweeps' view of this half
            if (Dh.pair_lens) Dh.pair_lens += b0 * P;
            Dh.corr24 = c24_it ? 1 : 0;
            Dh.corr_entry0 = (c24_it || relayout) ? (int64_t)b0 * corr_stride : 0;
            Dh.corr24_out = (relayout && it == 0) ? ws->corr24_tmp.as<float2>() : nullptr;
            if (Dh.block_ranges) Dh.block_ranges += b0 * N * (size_t)((Wd / 8) * (Hd / 8));
#ifdef BTBA_WG_TRACE
            static DevBuf wg_trace_buf;
            const char *wg_trace_file = std::getenv("BTBA_WG_TRACE_FILE");
            const size_t wg_trace_n = (size_t)tiles * D.n_dense_pairs * H.nb + (size_t)chunks * P * H.nb;
            if (wg_trace_file && it == prm->n_gn_iters - 1 && h == 0) {
                if ((rc = wg_trace_buf.ensure(32 * wg_trace_n))) return rc;
                HIP_TRY(hipMemsetAsync(wg_trace_buf.p, 0, 32 * wg_trace_n, H.st));
                Dh.wg_trace = wg_trace_buf.as<unsigned long long>();
            }
#endif
            const unsigned n_d = (unsigned)tiles * D.n_dense_pairs * H.nb, n_s = (unsigned)chunks * P * H.nb;
            // one interleaved launch of both sweeps: the dense workgroups are VALU-bound, the sparse ones stream HBM, and the
            // two fill each other's idle pipes -- measured at c3 (scripts/ab_dense.py, fused vs separate step time):
            // B=1 0.486 / 0.554 ms, B=8 0.871 / 0.950 ms, B=32 2.271 / 2.380 ms
            const bool fuse = use_sparse && use_dense_it && n_s >= 64 && n_d >= 64 && !(prm->flags & BTBA_FLAG_NO_FUSE);
            Dh.corr_nt_from = ((fuse && use_dense_it) || ws->tune.corr_nt >= 0) ? corr_nt_from - (int)b0 : B;      // (b counts from the group's first instance)
            if (fuse) {
                // one launch: HBM-streaming sparse workgroups interleaved with the VALU-bound dense ones
                if ((rc = time_begin(ws, timing_it, 0, &slot, H.st))) return rc;
#define BTBA_FUSED_ARGS(CACHE) Dh, n_d, n_s, CACHE, reinterpret_cast<const float4 *>(normals_h), ws->dense_pairs.as<int2>(), T_h, Ti_h, dp_h, corr_h, off_h, sp_h, vl_h, vc_h
                const int lay = zn_layout ? zn_layout + (compaction ? 2 : 0) : 0;
                if (lay == 0) k_fused_sweeps<0><<<dim3(n_d + n_s), kBlock, 0, H.st>>>(BTBA_FUSED_ARGS(reinterpret_cast<const float4 *>(campos_h)));
                else if (lay == 1) k_fused_sweeps<1><<<dim3(n_d + n_s), kBlock, lut_bytes, H.st>>>(BTBA_FUSED_ARGS(zn_h));
                else if (lay == 2) k_fused_sweeps<2><<<dim3(n_d + n_s), kBlock, lut_bytes, H.st>>>(BTBA_FUSED_ARGS(zn_h));
                else if (lay == 3) k_fused_sweeps<3><<<dim3(n_d + n_s), kBlock, lut_bytes, H.st>>>(BTBA_FUSED_ARGS(zn_h));
                else k_fused_sweeps<4><<<dim3(n_d + n_s), kBlock, lut_bytes, H.st>>>(BTBA_FUSED_ARGS(zn_h));
#undef BTBA_FUSED_ARGS
                if ((rc = time_end(ws, slot, H.st))) return rc;
            } else {
                if (use_sparse) {
                    if ((rc = time_begin(ws, timing_it, 1, &slot, H.st))) return rc;
                    k_sparse_sweep<<<dim3(chunks, P, H.nb), kBlock, 0, H.st>>>(Dh, corr_h, off_h, T_h, sp_h);
                    if ((rc = time_end(ws, slot, H.st))) return rc;
                }
                if (use_dense_it) {
                    if ((rc = time_begin(ws, timing_it, 0, &slot, H.st))) return rc;
                    const dim3 dgrid(n_d);
#define BTBA_DENSE_ARGS Di, reinterpret_cast<const float4 *>(campos_h), reinterpret_cast<const float4 *>(normals_h), ws->dense_pairs.as<int2>(), T_h, Ti_h, dp_h
                    if (zn_layout == 1 && !compaction) k_dense_sweep_zn<true, false><<<dgrid, kBlock, lut_bytes, H.st>>>(Dh, zn_h, ws->dense_pairs.as<int2>(), T_h, Ti_h, dp_h, vl_h, vc_h);
                    else if (zn_layout == 2 && !compaction) k_dense_sweep_zn<false, false><<<dgrid, kBlock, lut_bytes, H.st>>>(Dh, zn_h, ws->dense_pairs.as<int2>(), T_h, Ti_h, dp_h, vl_h, vc_h);
                    else if (zn_layout == 1) k_dense_sweep_zn<true, true><<<dgrid, kBlock, lut_bytes, H.st>>>(Dh, zn_h, ws->dense_pairs.as<int2>(), T_h, Ti_h, dp_h, vl_h, vc_h);
                    else if (zn_layout == 2) k_dense_sweep_zn<false, true><<<dgrid, kBlock, lut_bytes, H.st>>>(Dh, zn_h, ws->dense_pairs.as<int2>(), T_h, Ti_h, dp_h, vl_h, vc_h);
                    else k_dense_sweep<<<dgrid, kBlock, 0, H.st>>>(BTBA_DENSE_ARGS);
#undef BTBA_DENSE_ARGS
                    if ((rc = time_end(ws, slot, H.st))) return rc;
                }
            }
#ifdef BTBA_WG_TRACE
            if (Dh.wg_trace) {                                   // dump the last iteration's workgroup timeline
                std::vector<unsigned long long> host(4 * wg_trace_n);
                HIP_TRY(hipStreamSynchronize(H.st));
                HIP_TRY(hipMemcpy(host.data(), wg_trace_buf.p, 32 * wg_trace_n, hipMemcpyDeviceToHost));
                if (FILE *f = std::fopen(wg_trace_file, "wb")) { std::fwrite(host.data(), 8, host.size(), f); std::fclose(f); }
            }
#endif
            if ((rc = time_begin(ws, timing_it, 2, &slot, H.st))) return rc;
            float *A_h = (a_global || D.pre_assembled) ? ws->big_A.as<float>() + b0 * (n + 2) * ld : nullptr;
            float *out_h = (it == prm->n_gn_iters - 1) ? poses + 16 * b0 * N : nullptr;     // the last iterate's matrices go straight to the caller's buffer
            SolveDims Ds = Di;                                  // the system solve's view: in atomic mode one record per pair
            if (atomic_sums) { Ds.sparse_chunks = 1; Ds.dense_tiles = 1; }
            if (D.pre_assembled) {
                const size_t n_sums = (size_t)P * kSparseVals + (size_t)D.n_dense_pairs * kDenseVals;
                k_big_reduce<<<dim3((unsigned)((n_sums + 255) / 256), (unsigned)H.nb), 256, 0, H.st>>>(Ds, sp_h, dp_h, ps_h);
                const BigTasks bt = big_tasks(N, P, (int)ld);
                k_big_assemble<<<dim3((bt.total + 255u) / 256u, (unsigned)H.nb), 256, 0, H.st>>>(Di, ps_h, ws->dense_pairs.as<int2>(), d_adj_off, d_adj, ws->solve_tab.as<int>(), A_h);
            }
#define BTBA_SOLVE(LP, AG) k_system_solve<LP, AG><<<H.nb, kSolveBlock, lds_bytes, H.st>>>(Ds, it, sp_h, dp_h, ws->dense_pairs.as<int2>(), d_adj_off, d_adj, x_h, T_h, Ti_h, ps_h, tr_h, A_h, out_h, ws->solve_tab.as<int>())
            if (small_solve || mid_solve) {
                // tracker-sized windows: k_solve_small (btba_solve_small.hpp); 22 ... 31 frames: k_solve_mid (btba_solve_mid.hpp)
                SmallSolveArgs Sa{};
                Sa.n_frames = N; Sa.n_pairs = P; Sa.n_dense_pairs = use_dense_it ? D.n_dense_pairs : 0;
                Sa.sparse_chunks = Ds.sparse_chunks; Sa.dense_tiles = Ds.dense_tiles; Sa.n_pcg = D.n_pcg; Sa.use_sparse = use_sparse ? 1 : 0;
                Sa.w_sparse = Di.w_sparse;
                Sa.sp_stride = D.sp_stride; Sa.dp_stride = D.dp_stride; Sa.pose_stride = D.pose_stride; Sa.x_stride = D.x_stride;
                Sa.iter = it;
                Sa.trace_record = D.trace_record; Sa.tr_x = D.tr_x; Sa.tr_T = D.tr_T; Sa.tr_rhs = D.tr_rhs; Sa.tr_prec = D.tr_prec; Sa.tr_pcg = D.tr_pcg;
                Sa.tr_delta = D.tr_delta; Sa.tr_dpair = D.tr_dpair; Sa.tr_A = D.tr_A; Sa.tr_clk = D.tr_clk; Sa.trace_instance = (int64_t)D.n_gn * D.trace_record;
                Sa.sparse_partials = sp_h; Sa.dense_partials = dp_h;
                Sa.adj_off = d_adj_off; Sa.adj = d_adj; Sa.cross = d_adj + 2 * (size_t)Pd; Sa.pair_ij = ws->solve_tab.as<int>(); Sa.entry_lut = ws->solve_tab.as<int>() + (((size_t)P + 288 + 3) & ~(size_t)3);
                Sa.x = x_h; Sa.T = T_h; Sa.Tinv = Ti_h; Sa.poses_out = out_h; Sa.trace = D.trace_on ? tr_h : nullptr;
                if (mid_solve) k_solve_mid<<<H.nb, kSmallBlock, mid_lds, H.st>>>(Sa);
                else if (small_cpl_v == 4) k_solve_small<4><<<H.nb, kSmallBlock, small_lds, H.st>>>(Sa);
                else if (small_cpl_v == 8) k_solve_small<8><<<H.nb, kSmallBlock, small_lds, H.st>>>(Sa);
                else if (small_cpl_v == 12) k_solve_small<12><<<H.nb, kSmallBlock, small_lds, H.st>>>(Sa);
                else k_solve_small<16><<<H.nb, kSmallBlock, small_lds, H.st>>>(Sa);
            } else
            if (a_global) { if (D.pairsum_in_lds) BTBA_SOLVE(true, true); else BTBA_SOLVE(false, true); }
            else { if (D.pairsum_in_lds) BTBA_SOLVE(true, false); else BTBA_SOLVE(false, false); }
#undef BTBA_SOLVE
            if ((rc = time_end(ws, slot, H.st))) return rc;
        }
    }
    for (int g = 1; g < n_halves; g++) {
        HIP_TRY(hipEventRecord(ws->ev_join[g - 1], ws->aux_streams[g - 1]));
        HIP_TRY(hipStreamWaitEvent(ws->stream, ws->ev_join[g - 1], 0));
    }
    if ((rc = time_end(ws, reg))) return rc;
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

extern "C" {

int btba_collect_stats(btba_workspace *ws, btba_stats *stats)
{
    DeviceGuard device_guard(ws);
    if (!ws) return BTBA_EINVAL;
    HIP_TRY(hipStreamSynchronize(ws->stream));
    btba_stats &S = ws->stats;
    for (auto &ep : ws->events) {
        float ms = 0.0f;
        hipError_t e = hipEventElapsedTime(&ms, ep.a, ep.b);
        if (e == hipSuccess) {
            switch (ep.kind) {
            case 0: S.ms_dense_sweep += ms; S.n_dense_launches++; break;
            case 1: S.ms_sparse_sweep += ms; S.n_sparse_launches++; break;
            case 2: S.ms_system_solve += ms; S.n_solve_launches++; break;
            case 3: S.ms_solve += ms; break;
            case 4: S.ms_cache += ms; break;
            default: break;
            }
        }
        ws->event_pool.push_back(ep.a);
        ws->event_pool.push_back(ep.b);
    }
    ws->events.clear();
    if (stats) *stats = S;
    S.ms_dense_sweep = S.ms_sparse_sweep = S.ms_system_solve = S.ms_solve = S.ms_cache = 0.0f;
    S.n_dense_launches = S.n_sparse_launches = S.n_solve_launches = 0;
    return chain_check(ws);
}

int btba_solve_batch(btba_workspace *ws, const btba_params *params, int n_instances, int n_frames, int Hd, int Wd, const float *intr,
                     const float *campos_dev, const float *normals_dev, const btba_entryj *corr_dev, int64_t corr_stride,
                     const uint32_t *pair_offsets_dev, uint32_t max_corr_per_pair, const int32_t *dense_pairs, int n_dense_pairs,
                     float *poses_dev, float *trace_dev)
{
    DeviceGuard device_guard(ws);
    if (!ws) return BTBA_EINVAL;
    // Timed regions accumulate over calls until btba_collect_stats(); without BTBA_FLAG_TIME_KERNELS nothing
    // is recorded, so an un-collected caller does not grow the list.  A runaway list is recycled.
    if (ws->events.size() > 65536) {
        for (auto &ep : ws->events) { ws->event_pool.push_back(ep.a); ws->event_pool.push_back(ep.b); }
        ws->events.clear();
    }
    return solve_enqueue(ws, params, n_instances, n_frames, Hd, Wd, intr, campos_dev, normals_dev, ZnSpec{}, corr_dev, corr_stride,
                         pair_offsets_dev, max_corr_per_pair, dense_pairs, dense_pairs ? n_dense_pairs : -1, poses_dev, trace_dev);
}

int btba_solve_cached(btba_workspace *ws, const btba_params *params, int n_frames, int Hd, int Wd, const float *intr,
                      const float *campos_dev, const float *normals_dev, const btba_entryj *corr_dev, uint32_t n_corr,
                      const uint32_t *pair_offsets_dev, uint32_t max_corr_per_pair, const int32_t *dense_pairs, int n_dense_pairs,
                      float *poses_dev, float *trace_dev)
{
    DeviceGuard device_guard(ws);
    return btba_solve_batch(ws, params, 1, n_frames, Hd, Wd, intr, campos_dev, normals_dev, corr_dev, (int64_t)(n_corr ? n_corr : 1), pair_offsets_dev,
                            max_corr_per_pair, dense_pairs, n_dense_pairs, poses_dev, trace_dev);
}

static void scaled_intrinsics(int H, int W, int Hd, int Wd, const float *K, float intr[4], Mat4 *Kinv)
{
    // CUDACache.cpp:20-24
    intr[0] = K[0] * ((float)Wd / (float)W);
    intr[1] = K[4] * ((float)Hd / (float)H);
    intr[2] = K[2] * ((float)(Wd - 1) / (float)(W - 1));
    intr[3] = K[5] * ((float)(Hd - 1) / (float)(H - 1));
    // m_inputIntrinsicsInv (CUDACache.cpp:33): generic cofactor inverse of the 4x4 embedding of K, in fp32
    const float m[16] = { K[0], K[1], K[2], 0, K[3], K[4], K[5], 0, K[6], K[7], K[8], 0, 0, 0, 0, 1 };
    auto minor = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
        return m[4 * r0 + c0] * (m[4 * r1 + c1] * m[4 * r2 + c2] - m[4 * r1 + c2] * m[4 * r2 + c1])
             - m[4 * r0 + c1] * (m[4 * r1 + c0] * m[4 * r2 + c2] - m[4 * r1 + c2] * m[4 * r2 + c0])
             + m[4 * r0 + c2] * (m[4 * r1 + c0] * m[4 * r2 + c1] - m[4 * r1 + c1] * m[4 * r2 + c0]);
    };
    float adj[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            int rr[3], cc[3], a = 0, b = 0;
            for (int k = 0; k < 4; k++) { if (k != r) rr[a++] = k; if (k != c) cc[b++] = k; }
            float mn = minor(rr[0], rr[1], rr[2], cc[0], cc[1], cc[2]);
            adj[4 * c + r] = ((r + c) & 1) ? -mn : mn;
        }
    const float det = m[0] * adj[0] + m[1] * adj[4] + m[2] * adj[8] + m[3] * adj[12];
    const float rdet = 1.0f / det;
    for (int k = 0; k < 16; k++) Kinv->m[k] = adj[k] * rdet;
}

int btba_build_cache(btba_workspace *ws, int n_frames, int H, int W, const float *K, float image_downscale,
                     const float *const *depth_dev, const float *const *normal_dev,
                     float *campos_dev, float *normals_dev, int32_t *n_valid_dev, float *intr_out)
{
    DeviceGuard device_guard(ws);
    if (!ws || n_frames < 1 || H < 2 || W < 2 || !K || !depth_dev || !normal_dev || !campos_dev || !normals_dev || !(image_downscale >= 1.0f)) return BTBA_EINVAL;
    const int Wd = (int)(W / image_downscale), Hd = (int)(H / image_downscale);     // LossGPU.cu:56-57
    if (Wd < 2 || Hd < 2) return BTBA_EINVAL;
    float intr[4];
    Mat4 Kinv;
    scaled_intrinsics(H, W, Hd, Wd, K, intr, &Kinv);
    if (intr_out) std::memcpy(intr_out, intr, sizeof intr);
    int rc;
    if ((rc = ws->ptrs.ensure(sizeof(void *) * 2 * (size_t)n_frames))) return rc;
    std::vector<const float *> h(2 * (size_t)n_frames);
    for (int k = 0; k < n_frames; k++) {
        if (!depth_dev[k] || !normal_dev[k]) return BTBA_EINVAL;
        h[k] = depth_dev[k]; h[n_frames + k] = normal_dev[k];
    }
    HIP_TRY(hipMemcpyAsync(ws->ptrs.p, h.data(), sizeof(void *) * h.size(), hipMemcpyHostToDevice, ws->stream));
    HIP_TRY(hipStreamSynchronize(ws->stream));    // h is a local
    if (n_valid_dev) HIP_TRY(hipMemsetAsync(n_valid_dev, 0, sizeof(int32_t) * n_frames, ws->stream));
    const int npix = Wd * Hd;
    size_t slot;
    if (ws->events.size() > 4096) { for (auto &ep : ws->events) { ws->event_pool.push_back(ep.a); ws->event_pool.push_back(ep.b); } ws->events.clear(); }     // callers that never collect
    if ((rc = time_begin(ws, true, 4, &slot))) return rc;
    k_build_cache<<<dim3((npix + kBlock - 1) / kBlock, n_frames), kBlock, 0, ws->stream>>>(W, H, Wd, Hd, Kinv, ws->ptrs.as<const float *>(), ws->ptrs.as<const float *>() + n_frames,
                                                                                         reinterpret_cast<float4 *>(campos_dev), reinterpret_cast<float4 *>(normals_dev), n_valid_dev);
    if ((rc = time_end(ws, slot))) return rc;
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

static int pool_resolve(btba_workspace *ws, int N, int H, int W, int Hd, int Wd, const float *K, float downscale, const uint64_t *keys,
                        const float *const *depth_dev, const float *const *normal_dev, std::vector<int> &slot_of, int *n_built, const int32_t **nv_pinned);
static void pool_commit(btba_workspace *ws, int N, const std::vector<int> &slot_of, const int32_t *nv_pinned, std::vector<int32_t> &nv_out);
static int enqueue_frame_cache(btba_workspace *ws, int M, int H, int W, int Hd, int Wd, const float *const *ptrs_dev, const int *slots_dev, const int32_t *slots_host,
                               float4 *zn, int32_t *nvalid, uint32_t *lists, int *counts, float2 *ranges);

static int optimize_frames_impl(btba_workspace *ws_in, const btba_params *params_in, int n_frames, int H, int W, const float *K,
                         const btba_entryj *corres_host, uint32_t n_corres, const int *n_match_per_pair,
                         const float *const *depth_dev, const float *const *normal_dev, const uint64_t *frame_keys,
                         const int32_t *dense_pairs, int n_dense_pairs, float *poses, btba_stats *stats)
{
    (void)n_match_per_pair;   // stored and never read by the reference either (SBA.cpp:85)
    const auto t0 = std::chrono::steady_clock::now();
    btba_params prm;
    if (params_in) prm = *params_in; else btba_params_default(&prm);
    if (n_frames < 2 || H < 2 || W < 2 || !K || !depth_dev || !normal_dev || !poses || (!corres_host && n_corres)) return BTBA_EINVAL;
    if (prm.pair_policy == BTBA_PAIRS_EXPLICIT && (!dense_pairs || n_dense_pairs < 0)) return BTBA_EINVAL;
    btba_workspace *ws = ws_in;
    int rc = BTBA_OK;
    if (!ws) { if ((rc = btba_workspace_create_on_stream(&ws, nullptr))) return rc; }       // the reference's stream: the legacy NULL stream
    // an error return may leave asynchronous copies out of this function's locals (offsets, poses, descriptors) in flight: drain them
    auto finish = [&](int code) { if (code != BTBA_OK && ws_in) { if (ws->copy_stream) (void)hipStreamSynchronize(ws->copy_stream); (void)hipStreamSynchronize(ws->stream); } if (!ws_in) btba_workspace_destroy(ws); return code; };
    for (auto &ep : ws->events) { ws->event_pool.push_back(ep.a); ws->event_pool.push_back(ep.b); }
    ws->events.clear();

    const int N = n_frames, P = N * (N - 1) / 2;
    const int Wd = (int)(W / prm.image_downscale), Hd = (int)(H / prm.image_downscale);
    if (Wd < 2 || Hd < 2) return finish(BTBA_EINVAL);
    const int npix = Wd * Hd;

    // A0/A6: correspondences by frame pair.  Bundler::optimizeGPU hands the array over pair-major with the segment lengths
    // in n_match_per_pair (Bundler.cpp:298-323): when the lengths add up, the array is uploaded as it is and the sparse
    // sweep itself checks every entry against the pair of its segment (a device flag read back with the poses); only
    // if that check fails -- or without n_match_per_pair -- the host buckets the array (a pass over all of it).
    std::vector<uint32_t> offsets(P + 1, 0);
    std::unique_ptr<btba_entryj[]> scattered;                 // only when the caller's order is not pair-major already
    const btba_entryj *upload = corres_host;
    uint32_t kept = 0, max_per_pair = 0;
    bool trust = false;
    if (n_match_per_pair && n_corres) {
        uint64_t tot = 0;
        bool ok = true;
        for (int p = 0; p < P && ok; p++) {
            ok = n_match_per_pair[p] >= 0;
            tot += (uint64_t)(ok ? n_match_per_pair[p] : 0);
            offsets[p + 1] = (uint32_t)tot;
        }
        trust = ok && tot == n_corres;
    }
    auto bucket_on_host = [&]() -> int {
        bool bucketed = false;
        std::fill(offsets.begin(), offsets.end(), 0u);
        if (int r = count_correspondences(corres_host, n_corres, N, offsets.data(), &bucketed)) return r;
        kept = offsets[P];
        upload = corres_host;
        if (!bucketed && kept) {
            scattered.reset(new btba_entryj[kept]);
            scatter_correspondences(corres_host, n_corres, N, offsets.data(), scattered.get());
            upload = scattered.get();
        }
        return BTBA_OK;
    };
    if (trust) kept = n_corres;
    else if ((rc = bucket_on_host())) return finish(rc);
    auto longest_segment = [&]() { uint32_t m = 0; for (int p = 0; p < P; p++) m = std::max(m, offsets[p + 1] - offsets[p]); return m; };
    max_per_pair = longest_segment();

    if ((rc = ws->corr.ensure(sizeof(btba_entryj) * (size_t)(kept ? kept : 1)))) return finish(rc);
    if ((rc = ws->offsets.ensure(sizeof(uint32_t) * (P + 1)))) return finish(rc);
    if ((rc = ws->poses.ensure(sizeof(float) * (16 * (size_t)N + 1)))) return finish(rc);      // + the order flag
    if ((rc = ws->campos.ensure(sizeof(float) * 4 * (size_t)N * npix))) return finish(rc);
    if ((rc = ws->normals.ensure(sizeof(float) * 4 * (size_t)N * npix))) return finish(rc);
    if ((rc = ws->nvalid.ensure(sizeof(int32_t) * N))) return finish(rc);
    auto hip_fail = [&](hipError_t e) { g_last_hip_error = (int)e; return finish(BTBA_EHIP); };
    hipError_t e;
    // poses in (+ one word for the device's "not pair-major" flag, 0), poses out and the pair offsets go through a pinned block of the workspace (round 6)
    const size_t io_floats = 16 * (size_t)N + 1;
    if ((rc = ws->pin_io_ensure(sizeof(float) * 2 * io_floats + sizeof(uint32_t) * (size_t)(P + 1) + sizeof(uint32_t) * 5 * (size_t)P + 64))) return finish(rc);
    float *out = static_cast<float *>(ws->pin_io), *stage = out + io_floats;
    uint32_t *offsets_pin = reinterpret_cast<uint32_t *>(stage + io_floats), *lens_pin = offsets_pin + (P + 1), *desc_pin = lens_pin + P;      // (keyed correspondences: segment lengths [P], descriptors [<= 4 P])
    // Keyed correspondence cache: with frame keys, the trusted pair-major layout and BTBA_FLAG_KEYED_CORR, a pair's segment is looked
    // up under (key_i, key_j, count); only the segments not seen before cross PCIe (into the pool), then one small kernel gathers
    // the window's P segments from the pool into the contiguous pair-major array the sweeps read, rewriting imgIdx_i / imgIdx_j to
    // the frames' CURRENT window positions (they shift as the window slides) and checking the new segments' own indices.
    int corr_pairs_uploaded = P;
    uint32_t stage_longest_fresh = 0;
    std::vector<uint32_t> offsets_up;                           // the offsets array as uploaded (pool offsets on the keyed path)
    // (below 1 MB the whole array crosses PCIe faster than the bookkeeping runs)
    const size_t kc_min = ws->tune.keyed_corr_min_bytes;                  // tests lower the threshold (BTBA_OPT_KEYED_CORR_MIN_BYTES)
    bool use_corr_cache = frame_keys != nullptr && trust && ws_in && (prm.flags & BTBA_FLAG_KEYED_CORR) && kept > 0 && (size_t)kept * sizeof(btba_entryj) >= kc_min;
    // Keyed path: the pool holds 24-BYTE correspondences (pos_i, pos_j: the frame indices are implied by the segment and would have to be
    // rewritten whenever the window slides); the sweeps read the window's segments where they lie in the pool (offset + length per pair),
    // nothing is gathered.  A fresh segment goes host -> pinned staging -> device staging (EntryJ) -> k_pack_corr24 -> pool, and that
    // kernel checks its entries against (i, j).  The index is committed only after every copy and launch of this call has been
    // enqueued: an error on the way leaves no entry behind that points at unwritten pool space.
    std::vector<uint32_t> desc, lens;                           // outlive the asynchronous copies (the call ends with a synchronisation)
    std::vector<std::pair<std::pair<uint64_t, uint64_t>, btba_workspace::CorrSeg>> fresh_index;
    size_t fresh_entries = 0;
    bool corr_in_pool = false;
    hipStream_t up_st = ws->stream;                             // the stream upload_inputs enqueues its copies on
    auto upload_inputs = [&]() -> hipError_t {
        hipError_t r;
        corr_in_pool = false;
        fresh_index.clear(); fresh_entries = 0;
        std::vector<uint32_t> offs_up(offsets);                 // what the sweeps get: pair-major offsets, or pool offsets + lens
        if (use_corr_cache && trust) {
            size_t need = 0;
            for (int pass = 0; pass < 2; pass++) {                  // pass 1 only after the pool had to be reset
                need = 0;
                int q = 0;
                for (int i = 0; i < N; i++)
                    for (int j = i + 1; j < N; j++, q++) {
                        const uint32_t cnt = offsets[q + 1] - offsets[q];
                        auto it = ws->corr_index.find({ frame_keys[i], frame_keys[j] });
                        if (cnt && (it == ws->corr_index.end() || it->second.count != cnt)) need += cnt;
                    }
                const size_t cap = ws->corr_pool.cap / 24 >= 64 ? ws->corr_pool.cap / 24 - 64 : 0;      // (the last 64-entry group may be partly used)
                if (ws->corr_pool_used + need <= cap) break;
                // does not fit: start over with an empty pool (sized for a few windows) -- this call then uploads everything once
                ws->corr_index.clear();
                ws->corr_pool_used = 0;
                if (ws->corr_pool.ensure(24 * (std::max<size_t>(16 * (size_t)kept, 1u << 16) + 64)) != BTBA_OK) return hipErrorOutOfMemory;      // ~16 windows' worth: resets are rare
            }
            corr_pairs_uploaded = 0;
            // the new segments are packed into ONE pinned staging buffer and cross PCIe in one copy (a pageable hipMemcpyAsync per
            // segment costs ~10 us each: 14 of them were as slow as uploading everything)
            if (need * sizeof(btba_entryj) > ws->corr_stage_cap) {
                if (ws->corr_stage) (void)hipHostFree(ws->corr_stage);
                ws->corr_stage = nullptr; ws->corr_stage_cap = 0;
                const size_t want = std::max<size_t>(2 * need * sizeof(btba_entryj), 1u << 20);
                if ((r = hipHostMalloc(&ws->corr_stage, want, hipHostMallocDefault)) != hipSuccess) return r;
                ws->corr_stage_cap = want;
            }
            if (ws->corr_stage_dev.ensure(sizeof(btba_entryj) * (need ? need : 1)) != BTBA_OK) return hipErrorOutOfMemory;
            const size_t pool_base = ws->corr_pool_used;
            size_t staged = 0;
            uint32_t longest_fresh = 0;
            desc.clear();
            lens.assign((size_t)P, 0u);
            int q = 0;
            for (int i = 0; i < N; i++)
                for (int j = i + 1; j < N; j++, q++) {
                    const uint32_t cnt = offsets[q + 1] - offsets[q];
                    uint32_t at = 0;
                    if (cnt) {
                        auto key = std::make_pair(frame_keys[i], frame_keys[j]);
                        auto it = ws->corr_index.find(key);
                        if (it == ws->corr_index.end() || it->second.count != cnt) {
                            at = (uint32_t)(pool_base + staged);
                            std::memcpy(static_cast<btba_entryj *>(ws->corr_stage) + staged, corres_host + offsets[q], sizeof(btba_entryj) * cnt);
                            desc.insert(desc.end(), { (uint32_t)staged, at, cnt, ((uint32_t)i << 16) | (uint32_t)j });      // staging offset -> pool offset
                            staged += cnt;
                            fresh_index.push_back({ key, btba_workspace::CorrSeg{ at, cnt } });
                            longest_fresh = std::max(longest_fresh, cnt);
                            corr_pairs_uploaded++;
                        } else at = it->second.off;
                    }
                    offs_up[q] = at;
                    lens[q] = cnt;
                }
            offs_up[P] = 0;
            fresh_entries = staged;
            if (staged) {
                if ((r = hipMemcpyAsync(ws->corr_stage_dev.p, ws->corr_stage, sizeof(btba_entryj) * staged, hipMemcpyHostToDevice, up_st)) != hipSuccess) return r;
                if (ws->corr_desc.ensure(sizeof(uint32_t) * desc.size()) != BTBA_OK) return hipErrorOutOfMemory;
                std::memcpy(desc_pin, desc.data(), sizeof(uint32_t) * desc.size());
                if ((r = hipMemcpyAsync(ws->corr_desc.p, desc_pin, sizeof(uint32_t) * desc.size(), hipMemcpyHostToDevice, up_st)) != hipSuccess) return r;
            }
            if (ws->corr_lens.ensure(sizeof(uint32_t) * (size_t)P) != BTBA_OK) return hipErrorOutOfMemory;
            std::memcpy(lens_pin, lens.data(), sizeof(uint32_t) * (size_t)P);
            if ((r = hipMemcpyAsync(ws->corr_lens.p, lens_pin, sizeof(uint32_t) * (size_t)P, hipMemcpyHostToDevice, up_st)) != hipSuccess) return r;
            corr_in_pool = true;
            stage_longest_fresh = longest_fresh;
        } else if (kept && (r = hipMemcpyAsync(ws->corr.p, upload, sizeof(btba_entryj) * kept, hipMemcpyHostToDevice, up_st)) != hipSuccess) return r;
        offsets_up.swap(offs_up);
        std::memcpy(offsets_pin, offsets_up.data(), sizeof(uint32_t) * (size_t)(P + 1));
        if ((r = hipMemcpyAsync(ws->offsets.p, offsets_pin, sizeof(uint32_t) * (P + 1), hipMemcpyHostToDevice, up_st)) != hipSuccess) return r;
        std::memcpy(stage, poses, sizeof(float) * 16 * N);
        stage[16 * (size_t)N] = 0.0f;
        return hipMemcpyAsync(ws->poses.p, stage, sizeof(float) * io_floats, hipMemcpyHostToDevice, up_st);
    };
    // ---- round 6: frame cache FIRST, on the workspace's stream; the EntryJ / pose upload meanwhile on a stream of its own (a pageable hipMemcpyAsync blocks
    // the host for the copy's ~0.13 ms at c3: the cache kernel runs under it); `stream` waits for the upload's event before anything reads the correspondences
    float intr[4];
    const bool compact = !(prm.flags & BTBA_FLAG_FLOAT4_CACHE);        // compact (z, n) cache: identical results, half the bytes
    const bool keyed = frame_keys != nullptr;
    if (keyed && (!compact || !ws_in)) return finish(BTBA_EINVAL);     // the persistent cache lives in a caller-owned workspace
    if (!ws->copy_stream) {
        if ((e = hipStreamCreateWithFlags(&ws->copy_stream, hipStreamNonBlocking)) != hipSuccess) return hip_fail(e);
        if ((e = hipEventCreateWithFlags(&ws->ev_copy, hipEventDisableTiming)) != hipSuccess) return hip_fail(e);
        if ((e = hipEventCreateWithFlags(&ws->ev_cache, hipEventDisableTiming)) != hipSuccess) return hip_fail(e);
    }
    std::vector<int32_t> nv_host;                               // valid pixels per frame of the window (known to the host once ev_cache has passed)
    std::vector<int> slot_of;
    const int32_t *nv_pinned = nullptr;
    int n_built = N;
    bool have_aux = false;                                      // stateless compact path: lists / counts / block ranges built with the cache (one launch)
    if (keyed) {
        Mat4 Kinv_unused;
        scaled_intrinsics(H, W, Hd, Wd, K, intr, &Kinv_unused);
        rc = pool_resolve(ws, N, H, W, Hd, Wd, K, prm.image_downscale, frame_keys, depth_dev, normal_dev, slot_of, &n_built, &nv_pinned);
    } else if (compact) {
        Mat4 Kinv_unused;
        scaled_intrinsics(H, W, Hd, Wd, K, intr, &Kinv_unused);
        const size_t off_nv = (sizeof(void *) * 2 * (size_t)N + 15) & ~(size_t)15;
        if ((rc = ws->pin_ensure(off_nv + sizeof(int32_t) * (size_t)N))) return finish(rc);
        if ((rc = ws->ptrs.ensure(off_nv))) return finish(rc);
        if ((rc = ws->valid_lists.ensure(sizeof(uint32_t) * (size_t)N * npix))) return finish(rc);
        if ((rc = ws->valid_counts.ensure(sizeof(int) * (size_t)N))) return finish(rc);
        if ((rc = ws->block_ranges.ensure(sizeof(float2) * (size_t)N * ((Wd / 8) * (Hd / 8) + 1)))) return finish(rc);
        auto **pp = reinterpret_cast<const float **>(ws->pin);
        for (int k = 0; k < N; k++) {
            if (!depth_dev[k] || !normal_dev[k]) return finish(BTBA_EINVAL);
            pp[k] = depth_dev[k]; pp[N + k] = normal_dev[k];
        }
        if ((e = hipMemcpyAsync(ws->ptrs.p, ws->pin, sizeof(void *) * 2 * (size_t)N, hipMemcpyHostToDevice, ws->stream)) != hipSuccess) return hip_fail(e);
        rc = enqueue_frame_cache(ws, N, H, W, Hd, Wd, ws->ptrs.as<const float *>(), nullptr, nullptr, ws->campos.as<float4>(), ws->nvalid.as<int32_t>(),
                                 ws->valid_lists.as<uint32_t>(), ws->valid_counts.as<int>(), ws->block_ranges.as<float2>());
        if (!rc) {
            nv_pinned = reinterpret_cast<const int32_t *>(static_cast<unsigned char *>(ws->pin) + off_nv);
            if ((e = hipMemcpyAsync(static_cast<unsigned char *>(ws->pin) + off_nv, ws->nvalid.p, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost, ws->stream)) != hipSuccess) return hip_fail(e);
            have_aux = true;
        }
    } else rc = btba_build_cache(ws, N, H, W, K, prm.image_downscale, depth_dev, normal_dev, ws->campos.as<float>(), ws->normals.as<float>(), ws->nvalid.as<int32_t>(), intr);
    if (rc) return finish(rc);
    if ((e = hipEventRecord(ws->ev_cache, ws->stream)) != hipSuccess) return hip_fail(e);

    const auto tu0 = std::chrono::steady_clock::now();
    up_st = ws->copy_stream;
    if ((e = upload_inputs()) != hipSuccess) return hip_fail(e);
    if ((e = hipEventRecord(ws->ev_copy, ws->copy_stream)) != hipSuccess) return hip_fail(e);
    if ((e = hipStreamWaitEvent(ws->stream, ws->ev_copy, 0)) != hipSuccess) return hip_fail(e);
    auto pack_fresh = [&]() -> hipError_t {                     // after upload_inputs: fresh segments staging -> pool (24 B), order check into the flag word
        if (!corr_in_pool) return hipSuccess;
        if (fresh_entries)
            k_pack_corr24<<<dim3((stage_longest_fresh + 255u) / 256u, (unsigned)(desc.size() / 4)), 256, 0, ws->stream>>>(
                ws->corr_desc.as<uint4>(), reinterpret_cast<const uint4 *>(ws->corr_stage_dev.p), reinterpret_cast<float2 *>(ws->corr_pool.p),
                reinterpret_cast<int *>(ws->poses.as<float>() + 16 * (size_t)N));
        hipError_t r = hipGetLastError();
        if (r != hipSuccess) return r;
        for (auto &kv : fresh_index) ws->corr_index[kv.first] = kv.second;      // committed: everything they point at has been enqueued
        ws->corr_pool_used += fresh_entries;
        return hipSuccess;
    };
    if ((e = pack_fresh()) != hipSuccess) return hip_fail(e);
    // the sources (caller's arrays, `offsets`, `scattered`, `stage`) outlive the synchronising end of this call; the sync only serves the upload timer
    if ((prm.flags & BTBA_FLAG_TIME_KERNELS) && (e = hipStreamSynchronize(ws->copy_stream)) != hipSuccess) return hip_fail(e);
    const auto tu1 = std::chrono::steady_clock::now();
    // the cache launch and the copy of its valid counts were enqueued before the upload started: long done
    if (nv_pinned) {
        if ((e = hipEventSynchronize(ws->ev_cache)) != hipSuccess) return hip_fail(e);
        if (keyed) pool_commit(ws, N, slot_of, nv_pinned, nv_host);
        else nv_host.assign(nv_pinned, nv_pinned + N);
    }

    std::vector<int32_t> pairs;
    const int32_t *pairs_ptr = nullptr;
    int n_pairs_dense = -1;
    if (prm.pair_policy == BTBA_PAIRS_EXPLICIT) {
        pairs_ptr = dense_pairs; n_pairs_dense = n_dense_pairs;
    } else if (prm.pair_policy == BTBA_PAIRS_TARGET_MORE_VALID) {
        std::vector<int32_t> nv(N);
        if (!nv_host.empty()) nv = nv_host;
        else {
            if ((e = hipMemcpyAsync(nv.data(), ws->nvalid.p, sizeof(int32_t) * N, hipMemcpyDeviceToHost, ws->stream)) != hipSuccess) return hip_fail(e);
            if ((e = hipStreamSynchronize(ws->stream)) != hipSuccess) return hip_fail(e);
        }
        for (int i = 0; i < N; i++)
            for (int j = i + 1; j < N; j++) {
                if (nv[i] >= nv[j]) { pairs.push_back(i); pairs.push_back(j); }   // ties: i<j (SolverBundling.cu:25-33)
                else { pairs.push_back(j); pairs.push_back(i); }
            }
        pairs_ptr = pairs.data(); n_pairs_dense = (int)pairs.size() / 2;
    } else if (prm.pair_policy == BTBA_PAIRS_TARGET_HIGHER) {
        for (int i = 0; i < N; i++) for (int j = i + 1; j < N; j++) { pairs.push_back(j); pairs.push_back(i); }      // target = j, source = i
        pairs_ptr = pairs.data(); n_pairs_dense = (int)pairs.size() / 2;
    } else if (prm.pair_policy != BTBA_PAIRS_TARGET_LOWER) {
        return finish(BTBA_EINVAL);
    }

    ZnSpec Z;
    if (compact) {
        Z.zn = keyed ? ws->pool_zn.as<float>() : ws->campos.as<float>(); Z.H = H; Z.W = W; Z.K = K;
        if (keyed) { Z.frame_slot = reinterpret_cast<const int *>(static_cast<const unsigned char *>(ws->pool_map.p) + ws->pool_map_offset); Z.lists = ws->pool_lists.as<uint32_t>(); Z.counts = ws->pool_counts.as<int>(); if (Wd % 8 == 0 && Hd % 8 == 0) Z.block_ranges = ws->pool_ranges.as<float>(); }
        else if (have_aux) { Z.lists = ws->valid_lists.as<uint32_t>(); Z.counts = ws->valid_counts.as<int>(); if (Wd % 8 == 0 && Hd % 8 == 0) Z.block_ranges = ws->block_ranges.as<float>(); }
        if (!(prm.flags & (BTBA_FLAG_COMPACTION | BTBA_FLAG_NO_COMPACTION))) {
            // a tracker's frames are masked to the object: walk valid-pixel lists when under 60 % of the pixels carry a depth.
            // (the cache builder counted them; this call is synchronous anyway, so the 4*N-byte read-back costs nothing extra)
            std::vector<int32_t> nv(N);
            if (!nv_host.empty()) nv = nv_host;
            else {
                if ((e = hipMemcpyAsync(nv.data(), ws->nvalid.p, sizeof(int32_t) * N, hipMemcpyDeviceToHost, ws->stream)) != hipSuccess) return hip_fail(e);
                if ((e = hipStreamSynchronize(ws->stream)) != hipSuccess) return hip_fail(e);
            }
            long tot = 0;
            for (int v : nv) tot += v;
            if (tot * 10 < (long)N * npix * 6) prm.flags |= BTBA_FLAG_COMPACTION;
        }
    }
    btba_stats S;
    int *order_flag = reinterpret_cast<int *>(ws->poses.as<float>() + 16 * (size_t)N);
    auto solve_and_read = [&]() -> int {
        int r = solve_enqueue(ws, &prm, 1, N, Hd, Wd, intr, compact ? nullptr : ws->campos.as<float>(), compact ? nullptr : ws->normals.as<float>(), Z,
                              corr_in_pool ? ws->corr_pool.as<btba_entryj>() : ws->corr.as<btba_entryj>(), (int64_t)(kept ? kept : 1),
                              ws->offsets.as<uint32_t>(), max_per_pair, pairs_ptr, n_pairs_dense, ws->poses.as<float>(), nullptr, (trust && !corr_in_pool) ? order_flag : nullptr,
                              corr_in_pool, corr_in_pool ? ws->corr_lens.as<uint32_t>() : nullptr);
        if (r) return r;
        hipError_t he;
        if ((he = hipMemcpyAsync(out, ws->poses.p, sizeof(float) * io_floats, hipMemcpyDeviceToHost, ws->stream)) != hipSuccess) { g_last_hip_error = (int)he; return BTBA_EHIP; }
        return btba_collect_stats(ws, &S);       // synchronises
    };
    ws->always_time_region = true;               // ms_solve is reported by this entry point whatever the flags
    rc = solve_and_read();
    if (!rc && trust) {
        uint32_t flag;
        std::memcpy(&flag, &out[16 * (size_t)N], sizeof flag);
        if (flag) {
            // the array was not pair-major after all: bucket it on the host and solve again from the caller's poses
            trust = false;
            if (use_corr_cache) { ws->corr_index.clear(); ws->corr_pool_used = 0; use_corr_cache = false; corr_pairs_uploaded = P; }
            if ((rc = bucket_on_host())) { ws->always_time_region = false; return finish(rc); }
            max_per_pair = longest_segment();
            if ((rc = ws->corr.ensure(sizeof(btba_entryj) * (size_t)(kept ? kept : 1)))) { ws->always_time_region = false; return finish(rc); }
            up_st = ws->stream;
            if ((e = upload_inputs()) != hipSuccess || (e = pack_fresh()) != hipSuccess) { ws->always_time_region = false; return hip_fail(e); }
            rc = solve_and_read();
        }
    }
    ws->always_time_region = false;
    if (rc) return finish(rc);
    for (size_t k = 0; k < 16 * (size_t)N; k++) if (!std::isfinite(out[k])) return finish(BTBA_ENUMERIC);
    std::memcpy(poses, out, sizeof(float) * 16 * (size_t)N);
    if (stats) {
        S.n_corr = kept;
        S.cache_frames_built = n_built;
        S.corr_pairs_uploaded = corr_pairs_uploaded;
        S.bytes_sparse_alg = (int64_t)32 * kept;
        S.ms_upload = std::chrono::duration<float, std::milli>(tu1 - tu0).count();
        S.ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        *stats = S;
    }
    return finish(BTBA_OK);
}

int btba_optimize_frames(btba_workspace *ws, const btba_params *params, int n_frames, int H, int W, const float *K,
                         const btba_entryj *corres_host, uint32_t n_corres, const int *n_match_per_pair,
                         const float *const *depth_dev, const float *const *normal_dev,
                         const int32_t *dense_pairs, int n_dense_pairs, float *poses, btba_stats *stats)
{
    DeviceGuard device_guard(ws);
    return optimize_frames_impl(ws, params, n_frames, H, W, K, corres_host, n_corres, n_match_per_pair, depth_dev, normal_dev, nullptr,
                                dense_pairs, n_dense_pairs, poses, stats);
}

int btba_optimize_frames_keyed(btba_workspace *ws, const btba_params *params, int n_frames, int H, int W, const float *K,
                               const btba_entryj *corres_host, uint32_t n_corres, const int *n_match_per_pair,
                               const float *const *depth_dev, const float *const *normal_dev, const uint64_t *frame_keys,
                               const int32_t *dense_pairs, int n_dense_pairs, float *poses, btba_stats *stats)
{
    DeviceGuard device_guard(ws);
    if (!ws || !frame_keys) return BTBA_EINVAL;
    return optimize_frames_impl(ws, params, n_frames, H, W, K, corres_host, n_corres, n_match_per_pair, depth_dev, normal_dev, frame_keys,
                                dense_pairs, n_dense_pairs, poses, stats);
}

int btba_frame_cache_clear(btba_workspace *ws)
{
    DeviceGuard device_guard(ws);
    if (!ws) return BTBA_EINVAL;
    for (auto &sl : ws->pool_slots) sl = btba_workspace::FrameSlot{};
    ws->corr_index.clear();
    ws->corr_pool_used = 0;
    return BTBA_OK;
}

int btba_frame_cache_evict(btba_workspace *ws, uint64_t frame_key)
{
    DeviceGuard device_guard(ws);
    if (!ws) return BTBA_EINVAL;
    for (auto &sl : ws->pool_slots) if (sl.live && sl.key == frame_key) sl = btba_workspace::FrameSlot{};
    for (auto it = ws->corr_index.begin(); it != ws->corr_index.end();)          // its correspondence segments go with it (the pool space is reclaimed at the next reset)
        it = (it->first.first == frame_key || it->first.second == frame_key) ? ws->corr_index.erase(it) : std::next(it);
    return BTBA_OK;
}

// Persistent frame cache: map every frame of this call to a pool slot, building only the frames that are not cached
// yet (same key AND same device buffers) -- in a tracker that is the one new frame, the keyframes were cached by
// earlier calls (the reference re-caches all K frames on every call, LossGPU.cu:74-78).  Least-recently-used slots
// are recycled.  Leaves the slot map on the device (ws->pool_map) and the per-frame valid counts in nv_out.
// The frame cache of M frames: compact cache, valid-pixel lists + counts, per-block depth ranges.  ptrs_dev: [M depth pointers][M normal pointers];
// slots_dev / slots_host: destination slot per frame or NULL (frame f -> slot f).  Enqueue only -- no host synchronisation.
static int enqueue_frame_cache(btba_workspace *ws, int M, int H, int W, int Hd, int Wd, const float *const *ptrs_dev, const int *slots_dev, const int32_t *slots_host,
                               float4 *zn, int32_t *nvalid, uint32_t *lists, int *counts, float2 *ranges)
{
    // Three launches back to back, nothing between them on the host: compact cache (all frames in one grid), ordered valid-pixel lists + counts, per-block depth
    // ranges.  (Round 6 tried them as ONE launch twice -- a workgroup per frame: 33 us per frame, its 1.5 MB of samples through one compute unit; sixteen slices
    // per frame with the last arriver finishing lists and ranges: 40 us for one frame, 143 us for fifteen -- against ~20 us for these three; since the call's
    // EntryJ upload now runs beside them on its own stream their time is hidden anyway.  profiles/r06/boundary_timing_experiments.json)
    const int npix = Hd * Wd;
    int rc;
    if (slots_host) { for (int m = 0; m < M; m++) HIP_TRY(hipMemsetAsync(nvalid + slots_host[m], 0, sizeof(int32_t), ws->stream)); }
    else HIP_TRY(hipMemsetAsync(nvalid, 0, sizeof(int32_t) * (size_t)M, ws->stream));
    size_t tslot;
    if ((rc = time_begin(ws, true, 4, &tslot))) return rc;
    k_build_cache_zn<<<dim3((npix + kBlock - 1) / kBlock, M), kBlock, 0, ws->stream>>>(W, H, Wd, Hd, ptrs_dev, ptrs_dev + M, zn, nvalid, slots_dev);
    k_valid_lists<<<M, 1024, 0, ws->stream>>>(npix, zn, lists, counts, slots_dev);
    if (Wd % 8 == 0 && Hd % 8 == 0)
        k_block_ranges<<<dim3((unsigned)(((Wd / 8) * (Hd / 8) + kBlock / 64 - 1) / (kBlock / 64)), (unsigned)M), kBlock, 0, ws->stream>>>(Wd, Hd, zn, slots_dev, ranges);
    if ((rc = time_end(ws, tslot))) return rc;
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

// Round 6: ENQUEUE ONLY -- no host synchronisation (rounds 3-5 synchronised twice in here, to keep two locals alive, and launched three kernels for the one new frame
// of a tracker's call: the keyed path lost to re-caching all fifteen frames, profiles/r05/boundary_timing.jsonl).  Everything the device reads from the host goes through
// the workspace's pinned block; the new frames' valid counts come back into it behind the launch; pool_commit() -- called once the caller has waited for that copy --
// marks the new slots live.  pin layout: [2 M pointers][M slots][N slot map][cap counts].
static int pool_resolve(btba_workspace *ws, int N, int H, int W, int Hd, int Wd, const float *K, float downscale, const uint64_t *keys,
                        const float *const *depth_dev, const float *const *normal_dev, std::vector<int> &slot_of, int *n_built, const int32_t **nv_pinned)
{
    const int npix = Hd * Wd;
    int rc;
    const bool same_geometry = ws->pool_H == H && ws->pool_W == W && ws->pool_npix == npix && ws->pool_downscale == downscale &&
                               std::memcmp(ws->pool_K, K, sizeof ws->pool_K) == 0;
    const size_t want_slots = std::max<size_t>(32, 2 * (size_t)N);
    if (!same_geometry || ws->pool_slots.size() < (size_t)N) {
        const size_t cap = std::max(want_slots, ws->pool_slots.size());
        if ((rc = ws->pool_zn.ensure(sizeof(float) * 4 * cap * npix))) return rc;
        if ((rc = ws->pool_lists.ensure(sizeof(uint32_t) * cap * npix))) return rc;
        if ((rc = ws->pool_counts.ensure(sizeof(int) * cap))) return rc;
        if ((rc = ws->pool_ranges.ensure(sizeof(float2) * cap * (size_t)((Wd / 8) * (Hd / 8) + 1)))) return rc;
        if ((rc = ws->pool_nvalid.ensure(sizeof(int32_t) * cap))) return rc;
        ws->pool_slots.assign(cap, btba_workspace::FrameSlot{});
        ws->pool_H = H; ws->pool_W = W; ws->pool_npix = npix; ws->pool_downscale = downscale;
        std::memcpy(ws->pool_K, K, sizeof ws->pool_K);
    }
    ws->pool_pending.clear();
    const size_t cap = ws->pool_slots.size();
    slot_of.assign(N, -1);
    std::vector<char> taken(cap, 0);
    for (int k = 0; k < N; k++) {
        if (!depth_dev[k] || !normal_dev[k]) return BTBA_EINVAL;
        for (int q = 0; q < k; q++) if (keys[q] == keys[k]) return BTBA_EINVAL;          // a frame appears once per window
        for (size_t sidx = 0; sidx < cap; sidx++) {
            const auto &sl = ws->pool_slots[sidx];
            if (sl.live && sl.key == keys[k] && sl.depth == depth_dev[k] && sl.normal == normal_dev[k]) { slot_of[k] = (int)sidx; taken[sidx] = 1; break; }
        }
    }
    std::vector<int> miss;
    for (int k = 0; k < N; k++) {
        if (slot_of[k] >= 0) { ws->pool_hits++; continue; }
        size_t best = cap;
        for (size_t sidx = 0; sidx < cap; sidx++) {
            if (taken[sidx]) continue;
            if (!ws->pool_slots[sidx].live) { best = sidx; break; }
            if (best == cap || ws->pool_slots[sidx].stamp < ws->pool_slots[best].stamp) best = sidx;
        }
        if (best == cap) return BTBA_ENOMEM;       // cannot happen: cap >= 2N
        slot_of[k] = (int)best; taken[best] = 1;
        ws->pool_slots[best].live = false;         // (being overwritten: live again at the commit)
        miss.push_back(k);
        ws->pool_misses++;
    }
    ws->pool_stamp++;
    for (int k = 0; k < N; k++) ws->pool_slots[slot_of[k]].stamp = ws->pool_stamp;
    *n_built = (int)miss.size();
    const int M = (int)miss.size();
    const size_t off_slots = sizeof(void *) * 2 * (size_t)M, off_map = off_slots + sizeof(int32_t) * (size_t)M, off_nv = (off_map + sizeof(int32_t) * (size_t)N + 15) & ~(size_t)15;
    if ((rc = ws->pin_ensure(off_nv + sizeof(int32_t) * cap))) return rc;
    unsigned char *pin = static_cast<unsigned char *>(ws->pin);
    int32_t *pin_map = reinterpret_cast<int32_t *>(pin + off_map);
    for (int k = 0; k < N; k++) pin_map[k] = slot_of[k];
    *nv_pinned = reinterpret_cast<const int32_t *>(pin + off_nv);
    // ONE copy out of the pinned block carries [pointers | destination slots | slot map] (ws->pool_map holds all three; Z.frame_slot points at its tail)
    const size_t up_bytes = off_map + sizeof(int32_t) * (size_t)N;
    if ((rc = ws->pool_map.ensure(up_bytes + 16))) return rc;
    ws->pool_map_offset = off_map;
    if (M > 0) {
        auto **pp = reinterpret_cast<const float **>(pin);
        auto *ps = reinterpret_cast<int32_t *>(pin + off_slots);
        for (int m = 0; m < M; m++) { pp[m] = depth_dev[miss[m]]; pp[M + m] = normal_dev[miss[m]]; ps[m] = slot_of[miss[m]]; }
        HIP_TRY(hipMemcpyAsync(ws->pool_map.p, pin, up_bytes, hipMemcpyHostToDevice, ws->stream));
        const int *slots_dev = reinterpret_cast<const int *>(reinterpret_cast<const unsigned char *>(ws->pool_map.p) + off_slots);
        if ((rc = enqueue_frame_cache(ws, M, H, W, Hd, Wd, ws->pool_map.as<const float *>(), slots_dev, ps, ws->pool_zn.as<float4>(), ws->pool_nvalid.as<int32_t>(),
                                      ws->pool_lists.as<uint32_t>(), ws->pool_counts.as<int>(), ws->pool_ranges.as<float2>()))) return rc;
        HIP_TRY(hipMemcpyAsync(pin + off_nv, ws->pool_nvalid.p, sizeof(int32_t) * cap, hipMemcpyDeviceToHost, ws->stream));
        for (int m = 0; m < M; m++) ws->pool_pending.push_back({ ps[m], keys[miss[m]], depth_dev[miss[m]], normal_dev[miss[m]] });
    } else HIP_TRY(hipMemcpyAsync(ws->pool_map.p, pin, up_bytes, hipMemcpyHostToDevice, ws->stream));      // (off_map == 0: the map alone)
    return BTBA_OK;
}

// After the caller has waited for the counts (ws->ev_cache): the frames cached by this call become live pool entries; nv_out = the window's valid counts.
static void pool_commit(btba_workspace *ws, int N, const std::vector<int> &slot_of, const int32_t *nv_pinned, std::vector<int32_t> &nv_out)
{
    for (const auto &pe : ws->pool_pending) {
        auto &sl = ws->pool_slots[pe.slot];
        sl.live = true; sl.key = pe.key; sl.depth = pe.depth; sl.normal = pe.normal; sl.n_valid = nv_pinned[pe.slot];
    }
    ws->pool_pending.clear();
    nv_out.resize(N);
    for (int k = 0; k < N; k++) nv_out[k] = ws->pool_slots[slot_of[k]].n_valid;
}

int btba_matrices_to_poses(btba_workspace *ws, int n, const float *T_dev, float *x_dev)
{
    DeviceGuard device_guard(ws);
    if (!ws || n < 1 || !T_dev || !x_dev) return BTBA_EINVAL;
    k_prepare<<<(n + 63) / 64, 64, 0, ws->stream>>>(n, T_dev, x_dev, nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

int btba_poses_to_matrices(btba_workspace *ws, int n, const float *x_dev, float *T_dev, float *Tinv_dev)
{
    DeviceGuard device_guard(ws);
    if (!ws || n < 1 || !x_dev || (!T_dev && !Tinv_dev)) return BTBA_EINVAL;
    k_poses_to_matrices<<<(n + 63) / 64, 64, 0, ws->stream>>>(n, x_dev, T_dev, Tinv_dev);
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

int btba_process_depth(btba_workspace *ws, int H, int W, const float *depth_in_dev, float *depth_out_dev,
                       int erode_radius, float erode_diff, float erode_ratio, int bf_radius, float sigma_d, float sigma_r)
{
    DeviceGuard device_guard(ws);
    if (!ws || H < 1 || W < 1 || !depth_in_dev || !depth_out_dev || depth_in_dev == depth_out_dev) return BTBA_EINVAL;
    if (erode_radius < 0 || bf_radius < 0 || erode_radius + 2 * bf_radius > 16 || !(sigma_d > 0.0f) || !(sigma_r > 0.0f)) return BTBA_EINVAL;
    DepthFilterParams P{ W, H, erode_radius, erode_diff, erode_ratio, bf_radius, sigma_d, sigma_r };
    const int h = erode_radius + 2 * bf_radius;
    const size_t lds = 2 * sizeof(float) * (size_t)(kTileW + 2 * h) * (kTileH + 2 * h);
    const dim3 grid((W + kTileW - 1) / kTileW, (H + kTileH - 1) / kTileH);
    if (erode_radius == 1 && bf_radius == 2) k_process_depth<1, 2><<<grid, 256, lds, ws->stream>>>(P, depth_in_dev, depth_out_dev);        // the tracker's stencils, unrolled
    else k_process_depth<-1, -1><<<grid, 256, lds, ws->stream>>>(P, depth_in_dev, depth_out_dev);
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

int btba_depth_to_normals(btba_workspace *ws, int H, int W, const float *K, const float *depth_dev, float *normals_dev, float *xyz_dev)
{
    DeviceGuard device_guard(ws);
    if (!ws || H < 1 || W < 1 || !K || !depth_dev || !normals_dev) return BTBA_EINVAL;
    float intr[4];
    Mat4 Kinv;
    scaled_intrinsics(H, W, H, W, K, intr, &Kinv);       // only the generic cofactor inverse of the 4x4 embedding is used
    k_depth_to_normals<<<dim3((W + 63) / 64, (H + 3) / 4), dim3(64, 4), 0, ws->stream>>>(W, H, Kinv, depth_dev, reinterpret_cast<float4 *>(normals_dev), reinterpret_cast<float4 *>(xyz_dev));
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

int btba_build_cache_zn(btba_workspace *ws, int n_frames, int H, int W, const float *K, float image_downscale,
                        const float *const *depth_dev, const float *const *normal_dev, float *zn_dev, int32_t *n_valid_dev, float *intr_out)
{
    DeviceGuard device_guard(ws);
    if (!ws || n_frames < 1 || H < 2 || W < 2 || !K || !depth_dev || !normal_dev || !zn_dev || !(image_downscale >= 1.0f)) return BTBA_EINVAL;
    const int Wd = (int)(W / image_downscale), Hd = (int)(H / image_downscale);
    if (Wd < 2 || Hd < 2) return BTBA_EINVAL;
    float intr[4];
    Mat4 Kinv;
    scaled_intrinsics(H, W, Hd, Wd, K, intr, &Kinv);
    if (intr_out) std::memcpy(intr_out, intr, sizeof intr);
    int rc;
    if ((rc = ws->ptrs.ensure(sizeof(void *) * 2 * (size_t)n_frames))) return rc;
    std::vector<const float *> h(2 * (size_t)n_frames);
    for (int k = 0; k < n_frames; k++) {
        if (!depth_dev[k] || !normal_dev[k]) return BTBA_EINVAL;
        h[k] = depth_dev[k]; h[n_frames + k] = normal_dev[k];
    }
    HIP_TRY(hipMemcpyAsync(ws->ptrs.p, h.data(), sizeof(void *) * h.size(), hipMemcpyHostToDevice, ws->stream));
    HIP_TRY(hipStreamSynchronize(ws->stream));    // h is a local
    if (n_valid_dev) HIP_TRY(hipMemsetAsync(n_valid_dev, 0, sizeof(int32_t) * n_frames, ws->stream));
    const int npix = Wd * Hd;
    size_t slot;
    if (ws->events.size() > 4096) { for (auto &ep : ws->events) { ws->event_pool.push_back(ep.a); ws->event_pool.push_back(ep.b); } ws->events.clear(); }
    if ((rc = time_begin(ws, true, 4, &slot))) return rc;
    k_build_cache_zn<<<dim3((npix + kBlock - 1) / kBlock, n_frames), kBlock, 0, ws->stream>>>(W, H, Wd, Hd, ws->ptrs.as<const float *>(), ws->ptrs.as<const float *>() + n_frames,
                                                                                            reinterpret_cast<float4 *>(zn_dev), n_valid_dev, nullptr);
    if ((rc = time_end(ws, slot))) return rc;
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

int btba_pack_zn(btba_workspace *ws, int64_t n_pixels_total, const float *campos_dev, const float *normals_dev, float *zn_dev)
{
    DeviceGuard device_guard(ws);
    if (!ws || n_pixels_total < 1 || !campos_dev || !normals_dev || !zn_dev) return BTBA_EINVAL;
    k_pack_zn<<<(unsigned)((n_pixels_total + kBlock - 1) / kBlock), kBlock, 0, ws->stream>>>((size_t)n_pixels_total, reinterpret_cast<const float4 *>(campos_dev),
                                                                                            reinterpret_cast<const float4 *>(normals_dev), reinterpret_cast<float4 *>(zn_dev));
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

int btba_zn_block_ranges(btba_workspace *ws, int n_frames_total, int Hd, int Wd, const float *zn_dev, float *ranges_dev)
{
    DeviceGuard device_guard(ws);
    if (!ws || n_frames_total < 1 || Hd < 8 || Wd < 8 || (Hd % 8) || (Wd % 8) || !zn_dev || !ranges_dev) return BTBA_EINVAL;
    const int nblk = (Wd / 8) * (Hd / 8);
    k_block_ranges<<<dim3((unsigned)((nblk + kBlock / 64 - 1) / (kBlock / 64)), (unsigned)n_frames_total), kBlock, 0, ws->stream>>>(Wd, Hd, reinterpret_cast<const float4 *>(zn_dev), nullptr, reinterpret_cast<float2 *>(ranges_dev));
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

int btba_zn_valid_lists(btba_workspace *ws, int n_frames_total, int Hd, int Wd, const float *zn_dev, uint32_t *lists_dev, int32_t *counts_dev)
{
    DeviceGuard device_guard(ws);
    if (!ws || n_frames_total < 1 || Hd < 2 || Wd < 2 || !zn_dev || !lists_dev || !counts_dev) return BTBA_EINVAL;
    k_valid_lists<<<n_frames_total, 1024, 0, ws->stream>>>(Hd * Wd, reinterpret_cast<const float4 *>(zn_dev), lists_dev, counts_dev, nullptr);
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

int btba_pack_correspondences24(btba_workspace *ws, int n_instances, int n_frames, const btba_entryj *corr_dev, int64_t corr_stride,
                                const uint32_t *pair_offsets_dev, uint32_t max_corr_per_pair, float *corr24_dev, int32_t *order_flag_dev)
{
    DeviceGuard device_guard(ws);
    if (!ws || n_instances < 1 || n_frames < 2 || n_frames > BTBA_MAX_FRAMES || !corr_dev || corr_stride < 1 || !pair_offsets_dev || !corr24_dev) return BTBA_EINVAL;
    if (max_corr_per_pair == 0) return BTBA_OK;
    const int P = n_frames * (n_frames - 1) / 2;
    k_pack_corr24_batch<<<dim3((max_corr_per_pair + 255u) / 256u, (unsigned)P, (unsigned)n_instances), 256, 0, ws->stream>>>(
        n_frames, P, corr_stride, pair_offsets_dev, reinterpret_cast<const uint4 *>(corr_dev), reinterpret_cast<float2 *>(corr24_dev), order_flag_dev);
    HIP_TRY(hipGetLastError());
    return BTBA_OK;
}

int btba_solve_batch_zn(btba_workspace *ws, const btba_params *params, int n_instances, int n_frames, int H, int W, const float *K,
                        const float *zn_dev, const btba_entryj *corr_dev, int64_t corr_stride,
                        const uint32_t *pair_offsets_dev, uint32_t max_corr_per_pair, const int32_t *dense_pairs, int n_dense_pairs,
                        float *poses_dev, float *trace_dev)
{
    DeviceGuard device_guard(ws);
    return btba_solve_batch_zn_aux(ws, params, n_instances, n_frames, H, W, K, zn_dev, nullptr, corr_dev, corr_stride, pair_offsets_dev, max_corr_per_pair,
                                   dense_pairs, n_dense_pairs, poses_dev, trace_dev);
}

int btba_solve_batch_zn_aux(btba_workspace *ws, const btba_params *params, int n_instances, int n_frames, int H, int W, const float *K,
                            const float *zn_dev, const btba_zn_aux *aux, const btba_entryj *corr_dev, int64_t corr_stride,
                            const uint32_t *pair_offsets_dev, uint32_t max_corr_per_pair, const int32_t *dense_pairs, int n_dense_pairs,
                            float *poses_dev, float *trace_dev)
{
    DeviceGuard device_guard(ws);
    if (!ws || !params || !K || !zn_dev || H < 2 || W < 2 || !(params->image_downscale >= 1.0f)) return BTBA_EINVAL;
    const int Wd = (int)(W / params->image_downscale), Hd = (int)(H / params->image_downscale);
    if (Wd < 2 || Hd < 2) return BTBA_EINVAL;
    float intr[4];
    Mat4 Kinv;
    scaled_intrinsics(H, W, Hd, Wd, K, intr, &Kinv);
    if (ws->events.size() > 65536) {
        for (auto &ep : ws->events) { ws->event_pool.push_back(ep.a); ws->event_pool.push_back(ep.b); }
        ws->events.clear();
    }
    ZnSpec Z;
    Z.zn = zn_dev; Z.H = H; Z.W = W; Z.K = K;
    if (aux) { Z.block_ranges = aux->block_ranges; if (aux->valid_lists && aux->valid_counts) { Z.lists = aux->valid_lists; Z.counts = aux->valid_counts; } }
    const bool c24 = aux && aux->corr24;
    return solve_enqueue(ws, params, n_instances, n_frames, Hd, Wd, intr, nullptr, nullptr, Z, c24 ? reinterpret_cast<const btba_entryj *>(aux->corr24) : corr_dev, corr_stride,
                         pair_offsets_dev, max_corr_per_pair, dense_pairs, dense_pairs ? n_dense_pairs : -1, poses_dev, trace_dev, nullptr, c24);
}


// ---- correspondence RANSAC ---------------------------------------------------------
int btba_ransac_pairs_ex(btba_workspace *ws, int hypothesis, int device_resident, int n_pairs, const float *ptsA, const float *ptsB, const int32_t *n_pts,
                         int n_trials, float dist_thres, const int32_t *samples, uint64_t seed,
                         int32_t *inlier_ids_out, int32_t *n_inliers_out, int32_t *best_trial_out, float *best_pose_out,
                         int32_t *trial_counts_out, float *trial_poses_out)
{
    DeviceGuard device_guard(ws);
    if (!ws || n_pairs < 1 || !n_pts || n_trials < 1 || !(dist_thres >= 0.0f) || !inlier_ids_out || !n_inliers_out) return BTBA_EINVAL;
    const bool draw_hash = (hypothesis & BTBA_RANSAC_DRAW_HASH) != 0;
    hypothesis &= ~BTBA_RANSAC_DRAW_HASH;
    if (hypothesis != BTBA_RANSAC_REFERENCE_SVD && hypothesis != BTBA_RANSAC_HORN) return BTBA_EINVAL;
    std::vector<int32_t> offsets(n_pairs + 1, 0);
    for (int p = 0; p < n_pairs; p++) {
        if (n_pts[p] < 0) return BTBA_EINVAL;
        offsets[p + 1] = offsets[p] + n_pts[p];
    }
    const size_t T = (size_t)offsets[n_pairs], NT = (size_t)n_pairs * n_trials;
    if (T && (!ptsA || !ptsB)) return BTBA_EINVAL;
    const bool dev = device_resident != 0;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t o = 0;
    const size_t o_a = o; o += dev ? 0 : al(16 * (T ? T : 1));
    const size_t o_b = o; o += dev ? 0 : al(16 * (T ? T : 1));
    const size_t o_off = o; o += al(4 * (size_t)(n_pairs + 1));
    const size_t o_smp = o; o += (samples && !dev) ? al(12 * NT) : al(4);
    const size_t o_pose = o; o += al(48 * NT);
    const size_t o_cnt = o; o += al(4 * NT);
    const size_t o_best = o; o += al(8 * (size_t)n_pairs);
    const size_t o_ids = o; o += dev ? 0 : al(4 * (T ? T : 1));
    const size_t o_nin = o; o += al(4 * (size_t)n_pairs);
    const size_t o_bt = o; o += al(4 * (size_t)n_pairs);
    const size_t o_bp = o; o += al(64 * (size_t)n_pairs);
    int rc;
    if ((rc = ws->ransac.ensure(o ? o : 256))) return rc;
    unsigned char *base = ws->ransac.as<unsigned char>();
    if (T && !dev) {
        HIP_TRY(hipMemcpyAsync(base + o_a, ptsA, 16 * T, hipMemcpyHostToDevice, ws->stream));
        HIP_TRY(hipMemcpyAsync(base + o_b, ptsB, 16 * T, hipMemcpyHostToDevice, ws->stream));
    }
    HIP_TRY(hipMemcpyAsync(base + o_off, offsets.data(), 4 * (size_t)(n_pairs + 1), hipMemcpyHostToDevice, ws->stream));
    HIP_TRY(hipStreamSynchronize(ws->stream));          // `offsets` is a local (16 B per pair: the only host wait of the device-resident form)
    if (samples && !dev) HIP_TRY(hipMemcpyAsync(base + o_smp, samples, 12 * NT, hipMemcpyHostToDevice, ws->stream));
    HIP_TRY(hipMemsetAsync(base + o_best, 0, 8 * (size_t)n_pairs, ws->stream));
    if (!samples && !draw_hash) {
        // the reference's per-trial cuRAND streams = one table of uniforms for all pairs; rebuilt only when the seed changes or
        // more trials are asked for than the table holds (a longer table for the same seed starts with the shorter one)
        if (ws->ransac_u_host.size() < 3 * (size_t)n_trials || ws->ransac_u_seed != seed) {
            HIP_TRY(hipStreamSynchronize(ws->stream));      // an earlier call's kernels / upload may still be using the old table
            ws->ransac_u_host.assign(3 * (size_t)n_trials, 0.0f);
            xorwow::ransac_uniform_table(seed, n_trials, ws->ransac_u_host.data());
            ws->ransac_u_seed = seed;
            if ((rc = ws->ransac_u.ensure(12 * (size_t)n_trials))) return rc;
            HIP_TRY(hipMemcpyAsync(ws->ransac_u.p, ws->ransac_u_host.data(), 12 * (size_t)n_trials, hipMemcpyHostToDevice, ws->stream));
        }
    }
    RansacDims D{};
    D.n_pairs = n_pairs; D.n_trials = n_trials; D.dist_thres = dist_thres; D.seed = seed; D.hypothesis = hypothesis;
    D.draw = samples ? 1 : (draw_hash ? 0 : 2);
    const float4 *dA = dev ? reinterpret_cast<const float4 *>(ptsA) : reinterpret_cast<const float4 *>(base + o_a);
    const float4 *dB = dev ? reinterpret_cast<const float4 *>(ptsB) : reinterpret_cast<const float4 *>(base + o_b);
    const int *dS = (samples && dev) ? samples : reinterpret_cast<const int *>(base + o_smp);
    // device-resident: results go straight to the caller's device buffers (the optional per-trial tables too)
    int *d_ids = dev ? inlier_ids_out : reinterpret_cast<int *>(base + o_ids);
    int *d_nin = dev ? n_inliers_out : reinterpret_cast<int *>(base + o_nin);
    int *d_bt = (dev && best_trial_out) ? best_trial_out : reinterpret_cast<int *>(base + o_bt);
    float *d_bp = (dev && best_pose_out) ? best_pose_out : reinterpret_cast<float *>(base + o_bp);
    int *d_cnt = (dev && trial_counts_out) ? trial_counts_out : reinterpret_cast<int *>(base + o_cnt);
    float *d_pose = (dev && trial_poses_out) ? trial_poses_out : reinterpret_cast<float *>(base + o_pose);
    k_ransac_vote<<<dim3((n_trials + 255) / 256, n_pairs), 256, 0, ws->stream>>>(
        D, dA, dB, reinterpret_cast<const int *>(base + o_off), dS, ws->ransac_u.as<float>(), d_pose, d_cnt, reinterpret_cast<unsigned long long *>(base + o_best));
    k_ransac_extract<<<n_pairs, 256, 0, ws->stream>>>(
        D, dA, dB, reinterpret_cast<const int *>(base + o_off), d_pose, reinterpret_cast<const unsigned long long *>(base + o_best), d_ids, d_nin, d_bt, d_bp);
    HIP_TRY(hipGetLastError());
    if (dev) return BTBA_OK;                          // asynchronous on the workspace stream
    // the inlier lists are written only up to each pair's count: fetch counts first, ids after
    HIP_TRY(hipMemcpyAsync(n_inliers_out, d_nin, 4 * (size_t)n_pairs, hipMemcpyDeviceToHost, ws->stream));
    if (T) HIP_TRY(hipMemcpyAsync(inlier_ids_out, d_ids, 4 * T, hipMemcpyDeviceToHost, ws->stream));
    if (best_trial_out) HIP_TRY(hipMemcpyAsync(best_trial_out, d_bt, 4 * (size_t)n_pairs, hipMemcpyDeviceToHost, ws->stream));
    if (best_pose_out) HIP_TRY(hipMemcpyAsync(best_pose_out, d_bp, 64 * (size_t)n_pairs, hipMemcpyDeviceToHost, ws->stream));
    if (trial_counts_out) HIP_TRY(hipMemcpyAsync(trial_counts_out, d_cnt, 4 * NT, hipMemcpyDeviceToHost, ws->stream));
    if (trial_poses_out) HIP_TRY(hipMemcpyAsync(trial_poses_out, d_pose, 48 * NT, hipMemcpyDeviceToHost, ws->stream));
    HIP_TRY(hipStreamSynchronize(ws->stream));
    return BTBA_OK;
}

int btba_ransac_pairs(btba_workspace *ws, int n_pairs, const float *ptsA_host, const float *ptsB_host, const int32_t *n_pts,
                      int n_trials, float dist_thres, const int32_t *samples_host, uint64_t seed,
                      int32_t *inlier_ids_out, int32_t *n_inliers_out, int32_t *best_trial_out, float *best_pose_out,
                      int32_t *trial_counts_out, float *trial_poses_out)
{
    DeviceGuard device_guard(ws);
    return btba_ransac_pairs_ex(ws, BTBA_RANSAC_REFERENCE_SVD, 0, n_pairs, ptsA_host, ptsB_host, n_pts, n_trials, dist_thres, samples_host, seed,
                                inlier_ids_out, n_inliers_out, best_trial_out, best_pose_out, trial_counts_out, trial_poses_out);
}

int btba_ransac_reference_uniforms(uint64_t seed, int n_trials, float *u_out)
{
    if (n_trials < 0 || (n_trials && !u_out)) return BTBA_EINVAL;
    xorwow::ransac_uniform_table(seed, n_trials, u_out);
    return BTBA_OK;
}

}  // extern "C"
