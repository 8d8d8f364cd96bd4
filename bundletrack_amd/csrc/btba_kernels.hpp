// btba_kernels.hpp -- the gfx950 kernels of the bundle-adjustment hot path.
//
// One Gauss-Newton iteration = three launches (reference: 37 launches, 11 memsets, one blocking
// D2H copy and a cudaMalloc/cudaFree pair):
//
//   sparse_sweep  (A11/A12)  one pass over EntryJ[]: per frame pair the Huber-weighted right-hand
//                            side, the Jacobi preconditioner diagonal AND the moment sums from
//                            which the *unweighted* sparse J^T J blocks follow in closed form.
//                            Replaces PCGInit_Kernel1 (one thread per frame!) and the 5 x
//                            (PCGStep_Kernel0 + PCGStep_Kernel1a) matrix-free passes.
//   dense_sweep   (A10)      the point-to-plane Jacobian sweep: per (pair, pixel) projective
//                            association, bilinear target lookup, residual, Huber weight and the
//                            closed-form row a = [-n_w ; n_w x w]; since row_i = -row_j only
//                            S = sum w a a^T (21) and g = sum w a res (6) are reduced per pair.
//                            Replaces BuildDenseSystem_Kernel + FlipJtJ_Kernel (and drops the dead
//                            FindDenseCorrespondences_Kernel twin pass).
//   system_solve  (A11-A13)  one workgroup per instance: fixed-order reduction of the partials,
//                            assembly of the 6N x 6N normal matrix in LDS, Jacobi-PCG entirely in
//                            LDS, SE(3) update, next iterate's T and T^-1.
//
// Summation is deterministic (fixed DPP tree inside a wave, fixed order across waves, tiles and
// pairs); the reference uses float atomics (SURVEY.md section 5 "race detection").
#pragma once
#ifndef BTBA_FUSED_WAVES
#define BTBA_FUSED_WAVES 6      // waves per SIMD the fused sweep is compiled for (80 VGPRs)
#endif
#if defined(BTBA_REFERENCE_ORDER) && !defined(BTBA_EXACT_DIV)
#define BTBA_EXACT_DIV 1        // the reference-order experiment build (tests/tools/reference_order_experiment.py) divides and takes roots exactly, like the oracle
#endif
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "btba_device.hpp"

namespace btba {

constexpr int kBlock = 256;          // 4 waves (sweeps)
constexpr int kSolveBlock = 1024;    // 16 waves: the per-instance solve is latency-bound serial phases, so go wide
constexpr int kSparseVals = 44;      // per-pair sparse partial record
constexpr int kDenseVals = 28;       // per-pair dense partial record: S(21) g(6) count(1)
constexpr float kEps = 0.000001f;    // FLOAT_EPSILON, SolverUtil.h:10
constexpr int kRedFloats = 4 * kSparseVals + 8;   // LDS scratch of a sweep workgroup: 4 waves x 44 sparse sums, or 4 x 28 dense sums + their total (28) + M (36)

// sparse record layout
//  0      n (valid count)
//  1..3   sum w_i           4..6   sum w_j
//  7..12  sum w_i w_i^T (xx xy xz yy yz zz)     13..18 same for w_j
//  19..27 sum w_i w_j^T (row-major 3x3)
//  28..30 sum rho r         31..33 sum rho (w_i x r)      34..36 sum rho (w_j x r)
//  37     sum rho
//  38..40 sum rho (wy^2+wz^2, wx^2+wz^2, wx^2+wy^2) at w_i     41..43 same at w_j

struct SolveDims {
    int n_frames;        // N
    int n_pairs;         // P = N(N-1)/2 canonical correspondence pairs
    int n_dense_pairs;   // Pd
    int npix, width, height;
    int sparse_chunks;   // workgroups per correspondence segment
    int dense_tiles;     // workgroups per dense pair
    int n_pcg;
    int use_sparse, use_dense;
    float fx, fy, cx, cy;
    float robust_delta, dist_thresh, normal_thresh, depth_min, depth_max;
    float dist2_thresh;  // dist_thresh^2 (fp32 product, formed once on the host)
    float wm1, hm1, wm2, hm2;     // (float)(width - 1), (height - 1), (width - 2), (height - 2): the dense sweep's clamp bounds -- from the host, so that they arrive in SCALAR registers
    float w_sparse, w_dense;
    int64_t corr_stride; // EntryJ per instance block
    int trace_on;
    int64_t trace_record; // floats per (instance, iteration) record
    int64_t tr_x, tr_T, tr_rhs, tr_prec, tr_pcg, tr_delta, tr_dpair, tr_A, tr_clk;
    int n_gn;
    int pairsum_in_lds;  // 1: stage reduced pair sums in LDS, 0: in global scratch
    int pre_assembled;   // large windows: k_big_reduce + k_big_assemble have built the system, k_system_solve only solves and updates
    int walk_blocks;     // pinhole sweep on the compact cache: waves walk 8 x 8 pixel blocks (width and height multiples of 8)
#ifdef BTBA_WG_TRACE
    unsigned long long *wg_trace; // developer build (scripts/wg_trace.py): per workgroup of the fused sweep (start, end) in 100 MHz ticks, hardware id, kind
#endif
    const int4 *dense_work;       // work position q -> (target, source, dense pair, -): the order the sweeps work the pairs off, heaviest first
    int work_formula;             // 1 / 2: the table is the closed form "all pairs by ascending |i - j|, then ascending i" with target = lower / higher
                                  // frame -- the pinhole sweeps then compute their item instead of loading it (one memory round trip less)
    const float2 *block_ranges;   // per (cache slot, 8 x 8 block): [min, max] valid depth (k_block_ranges); nullptr: no block is skipped
    int sparse_tail_256; // fused sweep: share (in 1/256) of the sparse items placed at the END of each XCD's item sequence instead of interleaved
    int tile_major;      // dense work order inside an instance: 1 = (tile, pair) -- all pairs' band t of the images together -- 0 = (pair, tile)
    int *order_flag;     // non-null: the sparse sweep ORs 1 into it when an entry does not belong to the pair of its segment
    int atomic_sums;     // BTBA_REDUCE_ATOMIC: sweep workgroups ADD their sums into one record per pair (float atomics) instead of storing one record
                         // per (pair, chunk / tile); k_system_solve reads those records (chunks = tiles = 1 on its side) and clears them for the next iteration
    int corr24;          // the correspondences are 24-byte (pos_i, pos_j) records (k_pack_corr24) instead of EntryJ
    int64_t corr_entry0; // 24-byte layout: entry index of this launch's first instance in the array
    const uint32_t *pair_lens;   // non-null: [B][P] segment lengths (segments not back to back: the keyed pool); else offsets[p + 1] - offsets[p]
    // compact (z, nx, ny, nz) frame cache: how a cached pixel maps back to camera space -- the arithmetic of k_build_cache
    float zn_ki[16];     // full-resolution intrinsicsInv (4x4 embedding, generic cofactor inverse)
    float zn_scale_w, zn_scale_h;   // (W-1)/(Wd-1), (H-1)/(Hd-1) of the nearest-neighbour resample
    int zn_simple;       // 1: zero skew and affine last row (ki[1]=ki[3]=ki[4]=ki[7]=ki[12]=ki[13]=ki[14]=0)
    // persistent frame cache: frame f of the solve lives in pool slot frame_slot[f] (nullptr: slot == f, contiguous cache)
    const int *frame_slot;
    // floats per INSTANCE in the pose arrays (T, Tinv: >= 16 N), the state array (x: >= 6 N) and the sweep partials.  The plain launches keep
    // the instances back to back; the chained launch (k_chain) pads every instance's region to whole 128-byte lines, so that no cache line is
    // shared by two regions that different workgroups publish at different times.
    int pose_stride, x_stride;
    unsigned sp_stride, dp_stride;      // (32-bit: an instance's partials are a few MB at most; 64-bit strides cost the short masked / sparse items ~100 scalar instructions)
    float2 *corr24_out;  // non-null (EntryJ sweeps only): the sweep also WRITES every entry it reads as a 24-byte record (corr24_index layout, entry index counted from
                         // corr_entry0 + b * corr_stride): the first iteration of a solve re-lays the caller's fresh EntryJ array out for the iterations that follow
    unsigned long long *live_blocks;   // non-null (BTBA_OPT_COUNT_LIVE, bench.py's roofline.executed): every block-walk workgroup adds the number of 8 x 8 blocks it walks
    int publish;         // k_chain: sweep workgroups store their partial records write-through (agent scope) -- another workgroup of the SAME launch reads them
    int corr_nt_from;    // instances b >= corr_nt_from read their correspondences with NON-TEMPORAL loads: once a batch's frames + correspondences exceed the memory-side
                         // cache, the read-once stream otherwise evicts the frames the dense items re-read every iteration (btba_api.hip: corr_nt_auto); what still fits
                         // beside the frames -- the first instances' correspondences -- stays cached, and below the limit plain loads are faster for everything
};

__device__ __forceinline__ size_t frame_slot_of(const SolveDims &D, size_t f) { return D.frame_slot ? (size_t)D.frame_slot[f] : f; }

// canonical pair index -> (i, j), i < j, outer i
__device__ __forceinline__ void pair_from_index(int p, int n, int &i, int &j)
{
    int ii = 0, rem = p;
    while (rem >= n - 1 - ii) { rem -= n - 1 - ii; ii++; }
    i = ii; j = ii + 1 + rem;
}
__device__ __forceinline__ int pair_index(int i, int j, int n) { return i * n - i * (i + 1) / 2 + (j - i - 1); }

__device__ __forceinline__ Mat4 load_mat4(const float *p)
{
    Mat4 m;
    const float4 *q = reinterpret_cast<const float4 *>(p);
    float4 a = q[0], b = q[1], c = q[2], d = q[3];
    m.m[0] = a.x; m.m[1] = a.y; m.m[2] = a.z; m.m[3] = a.w;
    m.m[4] = b.x; m.m[5] = b.y; m.m[6] = b.z; m.m[7] = b.w;
    m.m[8] = c.x; m.m[9] = c.y; m.m[10] = c.z; m.m[11] = c.w;
    m.m[12] = d.x; m.m[13] = d.y; m.m[14] = d.z; m.m[15] = d.w;
    return m;
}
// the same through a constant-address-space pointer (wave-uniform address: four s_load_dwordx4)
__device__ __forceinline__ Mat4 load_mat4_uniform(const float *p)
{
    Mat4 m;
    const float4 a = ld_const_f4(p), b = ld_const_f4(p + 4), c = ld_const_f4(p + 8), d = ld_const_f4(p + 12);
    m.m[0] = a.x; m.m[1] = a.y; m.m[2] = a.z; m.m[3] = a.w;
    m.m[4] = b.x; m.m[5] = b.y; m.m[6] = b.z; m.m[7] = b.w;
    m.m[8] = c.x; m.m[9] = c.y; m.m[10] = c.z; m.m[11] = c.w;
    m.m[12] = d.x; m.m[13] = d.y; m.m[14] = d.z; m.m[15] = d.w;
    return m;
}
__device__ __forceinline__ void store_mat4(float *p, const Mat4 &m)
{
    float4 *q = reinterpret_cast<float4 *>(p);
    q[0] = make_float4(m.m[0], m.m[1], m.m[2], m.m[3]);
    q[1] = make_float4(m.m[4], m.m[5], m.m[6], m.m[7]);
    q[2] = make_float4(m.m[8], m.m[9], m.m[10], m.m[11]);
    q[3] = make_float4(m.m[12], m.m[13], m.m[14], m.m[15]);
}

// ---- prepare: Log of the input matrices, then Exp and inverse (SBA.cu:71-79, SolverBundling.cu:890-897)
// keep_T (developer / experiment builds only, tests/tools/reference_order_experiment.py): the incoming matrix IS the first iterate's T -- no Exp(Log(.)) through
// the device's libm in between --, so that an experiment can start this path and the oracle from bit-identical matrices
__global__ void __launch_bounds__(64) k_prepare(int total, const float *__restrict__ poses, float *__restrict__ x, float *__restrict__ T, float *__restrict__ Tinv, int keep_T = 0)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const Mat4 M = load_mat4(poses + 16 * (size_t)idx);
    float rot[3], trans[3];
    matrix_to_pose(M, rot, trans);
    if (x) { float *o = x + 6 * (size_t)idx; o[0] = rot[0]; o[1] = rot[1]; o[2] = rot[2]; o[3] = trans[0]; o[4] = trans[1]; o[5] = trans[2]; }
    if (T || Tinv) {
        Mat4 E = pose_to_matrix(rot, trans);
#if defined(BTBA_DEV_EXPERIMENTS) || defined(BTBA_REFERENCE_ORDER)
        if (keep_T) E = M;
#endif
        if (T) store_mat4(T + 16 * (size_t)idx, E);
        if (Tinv) store_mat4(Tinv + 16 * (size_t)idx, mat_inverse(E));
    }
}

__global__ void __launch_bounds__(64) k_poses_to_matrices(int total, const float *__restrict__ x, float *__restrict__ T, float *__restrict__ Tinv)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float *o = x + 6 * (size_t)idx;
    const float rot[3] = { o[0], o[1], o[2] }, trans[3] = { o[3], o[4], o[5] };
    const Mat4 E = pose_to_matrix(rot, trans);
    if (T) store_mat4(T + 16 * (size_t)idx, E);
    if (Tinv) store_mat4(Tinv + 16 * (size_t)idx, mat_inverse(E));
}

// ---- frame cache (A3): CUDACache::storeFrame as ONE launch for all frames --------------------
// grid (ceil(npix/256), n_frames).  Reads only the full-res pixels the nearest-neighbour
// resample picks (CUDAImageUtil.cu:57-61) instead of converting the whole 640x480 image first.
__global__ void __launch_bounds__(kBlock) k_build_cache(int W, int H, int Wd, int Hd, Mat4 Kinv,
                                                       const float *const *__restrict__ depth, const float *const *__restrict__ normals,
                                                       float4 *__restrict__ campos_out, float4 *__restrict__ normals_out, int *__restrict__ n_valid)
{
#pragma clang fp contract(off)   // plain IEEE mul/add: the cache is bit-identical to the CPU oracle's
    const int f = blockIdx.y;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    const int npix = Wd * Hd;
    int valid = 0;
    if (o < npix) {
        const int x = o % Wd, y = o / Wd;
        const float scaleW = (float)(W - 1) / (float)(Wd - 1);
        const float scaleH = (float)(H - 1) / (float)(Hd - 1);
        const unsigned xi = (unsigned)(x * scaleW + 0.5f);
        const unsigned yi = (unsigned)(y * scaleH + 0.5f);
        if (xi < (unsigned)W && yi < (unsigned)H) {
            const size_t s = (size_t)yi * W + xi;
            const float d = depth[f][s];
            float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((double)d >= 0.1) {
                // intrinsicsInv * (x d, y d, d, d); output (x, y, w, 1)  (CUDAImageUtil.cu:310-327)
                const float vx = (float)xi * d, vy = (float)yi * d;
                cp.x = Kinv.m[0] * vx + Kinv.m[1] * vy + Kinv.m[2] * d + Kinv.m[3] * d;
                cp.y = Kinv.m[4] * vx + Kinv.m[5] * vy + Kinv.m[6] * d + Kinv.m[7] * d;
                cp.z = Kinv.m[12] * vx + Kinv.m[13] * vy + Kinv.m[14] * d + Kinv.m[15] * d;
                cp.w = 1.0f;
                valid = 1;
            }
            campos_out[(size_t)f * npix + o] = cp;
            normals_out[(size_t)f * npix + o] = reinterpret_cast<const float4 *>(normals[f])[s];
        }
    }
    if (n_valid) {
        const unsigned long long b = __ballot(valid);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(&n_valid[f], __popcll(b));
    }
}

// ---- compact frame cache ("ZN": float4 = depth z, normal x, y, z; 16 B per pixel instead of 32) -----------
// camPos is a pure function of (full-res pixel, depth): CUDAImageUtil.cu:310-327 computes
// intrinsicsInv * (x d, y d, d, d).  Storing only d and re-evaluating that expression with the SAME fp32
// operations (no contraction) where it is consumed gives bit-identical camera-space points while halving the
// bytes and the load instructions of the dense sweep (one 16-byte load per tap instead of two).
// Camera-space point of a cached pixel from its depth.  General intrinsics: the cache builder's arithmetic, term by
// term (cx, cy = (float) full-resolution column / row of the cached pixel).  SIMPLE (zero skew, affine last row -- every
// pinhole K): x = K^-1[0][0] (xi d) + K^-1[0][2] d = d (K^-1[0][0] xi + K^-1[0][2]) with the bracket tabulated per
// column (cx) / row (cy): three multiplies per point instead of nine operations; differs from the builder's rounding
// sequence by at most 1 ulp per coordinate.
template <bool SIMPLE>
__device__ __forceinline__ float3 zn_backproject(const float *ki, float cx, float cy, float d)
{
#pragma clang fp contract(off)
    // (double)d >= 0.1  <=>  d >= 0.1f  (0.1f is the smallest float above 0.1); below it the reference stores zeros,
    // and with d := 0 every product below is an exact zero as well, so one select replaces three.
    d = (d >= 0.1f) ? d : 0.0f;
    if (SIMPLE) return make_float3(cx * d, cy * d, ki[15] * d);
    const float vx = cx * d, vy = cy * d;
    return make_float3(ki[0] * vx + ki[1] * vy + ki[2] * d + ki[3] * d, ki[4] * vx + ki[5] * vy + ki[6] * d + ki[7] * d,
                       ki[12] * vx + ki[13] * vy + ki[14] * d + ki[15] * d);
}
__device__ __forceinline__ unsigned zn_src_coord(int c, float scale)
{
#pragma clang fp contract(off)
    return (unsigned)(c * scale + 0.5f);          // CUDAImageUtil.cu:60-61
}

// grid (ceil(npix/256), n_frames): CUDACache::storeFrame for all frames, compact output.
__global__ void __launch_bounds__(kBlock) k_build_cache_zn(int W, int H, int Wd, int Hd, const float *const *__restrict__ depth, const float *const *__restrict__ normals,
                                                          float4 *__restrict__ zn_out, int *__restrict__ n_valid, const int *__restrict__ out_slot)
{
#pragma clang fp contract(off)
    const int f = blockIdx.y;
    const int fo = out_slot ? out_slot[f] : f;        // where frame f is stored (persistent cache: a pool slot)
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    const int npix = Wd * Hd;
    int valid = 0;
    if (o < npix) {
        const int x = o % Wd, y = o / Wd;
        const float scaleW = (float)(W - 1) / (float)(Wd - 1);
        const float scaleH = (float)(H - 1) / (float)(Hd - 1);
        const unsigned xi = (unsigned)(x * scaleW + 0.5f);
        const unsigned yi = (unsigned)(y * scaleH + 0.5f);
        if (xi < (unsigned)W && yi < (unsigned)H) {
            const size_t s = (size_t)yi * W + xi;
            const float d = depth[f][s];
            const float4 nr = reinterpret_cast<const float4 *>(normals[f])[s];
            valid = ((double)d >= 0.1) ? 1 : 0;
            // GATED depth: 0 where the reference's camPos is (0, 0, 0, 0) (CUDAImageUtil.cu:310-327: d < 0.1, and NaN fails the
            // comparison too), so that the sweep blends exactly what the reference blends without re-testing every tap
            zn_out[(size_t)fo * npix + o] = make_float4(valid ? d : 0.0f, nr.x, nr.y, nr.z);
        }
    }
    if (n_valid) {
        const unsigned long long b = __ballot(valid);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(&n_valid[fo], __popcll(b));
    }
}

// float4 camPos + float4 normal caches -> compact cache (for callers that already hold the reference layout;
// camPos.xy are dropped and later recomputed from z, so the input must come from the standard formula)
__global__ void __launch_bounds__(kBlock) k_pack_zn(size_t total, const float4 *__restrict__ campos, const float4 *__restrict__ normals, float4 *__restrict__ zn)
{
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const float4 c = campos[o], nr = normals[o];
    zn[o] = make_float4(c.z, nr.x, nr.y, nr.z);
}

// ---- sparse sweep -----------------------------------------------------------------------------
// grid (sparse_chunks, P, B).  Workgroup (c, p, b) owns slice c of pair p's contiguous correspondence segment, 44 register
// accumulators per lane.  Two device layouts of a correspondence:
//   EntryJ (32 B, the wire format, SIFTImageManager.h:44-59): two coalesced 16-byte loads; the frame indices are checked against the
//            segment's pair on the fly (order_flag) -- what a call that uploads the caller's array runs on;
//   C24    (24 B: pos_i, pos_j; invalid <=> first word 0xFFFFFFFF): the indices are implied by the segment, so device-RESIDENT
//            correspondences (batches, the keyed pool) drop them -- a quarter of the stream that bounds the masked / feature-only launches
//            (k_pack_corr24 converts and checks the order once).  Stored in groups of 64 entries as three planes of float2 --
//            (pos_i.x, pos_i.y)[64], (pos_i.z, pos_j.x)[64], (pos_j.y, pos_j.z)[64] -- so that each of a wave's three 8-byte loads covers 512
//            contiguous bytes: plain 24-byte records, three loads at stride 24, touch every cache line three times and measured SLOWER than
//            EntryJ on the masked launch although they move a quarter less (52.0 vs 49.9 us, gpurun_out/r03_13).
// float2 index of entry E's first plane (the other two follow at + 64 and + 128); E counts entries from the start of the array
__device__ __forceinline__ size_t corr24_index(size_t E) { return (E >> 6) * 192 + (E & 63); }
template <bool C24, bool NT>
__device__ __forceinline__ void sparse_block_impl(const SolveDims &D, const float4 *__restrict__ corr, const uint32_t *__restrict__ pair_offsets,
                                                  const float *__restrict__ T, float *__restrict__ partials, int chunk, int p, int b, float *red)
{
    const unsigned tid = item_tid();
    int fi, fj;
    pair_from_index(p, D.n_frames, fi, fj);
    const auto off = as_const(pair_offsets + (size_t)b * (D.n_pairs + 1));
    const uint32_t seg0 = off[p];
    const uint32_t len = D.pair_lens ? as_const(D.pair_lens)[(size_t)b * D.n_pairs + p] : off[p + 1] - seg0;       // (pool segments are not back to back)
    const uint32_t per = (len + D.sparse_chunks - 1) / D.sparse_chunks;
    const uint32_t lo = seg0 + min(len, per * chunk), hi = seg0 + min(len, per * (chunk + 1));
    const float *Tb = T + (unsigned)b * (unsigned)D.pose_stride;
    const Mat4 Ti = load_mat4_uniform(Tb + 16 * fi);
    const Mat4 Tj = load_mat4_uniform(Tb + 16 * fj);
    float acc[kSparseVals];
#pragma unroll
    for (int k = 0; k < kSparseVals; k++) acc[k] = 0.0f;

    const float delta2 = D.robust_delta * D.robust_delta;
    auto accumulate = [&](bool valid, float pix, float piy, float piz, float pjx, float pjy, float pjz) {
        const float m = valid ? 1.0f : 0.0f;
        float wix, wiy, wiz, wjx, wjy, wjz;
        xform_point(Ti, pix, piy, piz, wix, wiy, wiz);
        xform_point(Tj, pjx, pjy, pjz, wjx, wjy, wjz);
        // an invalid / out-of-range slot contributes exact zeros (its payload may be anything, even NaN)
        wix = m != 0.0f ? wix : 0.0f; wiy = m != 0.0f ? wiy : 0.0f; wiz = m != 0.0f ? wiz : 0.0f;
        wjx = m != 0.0f ? wjx : 0.0f; wjy = m != 0.0f ? wjy : 0.0f; wjz = m != 0.0f ? wjz : 0.0f;
        const float rx = wix - wjx, ry = wiy - wjy, rz = wiz - wjz;
        const float e2 = rx * rx + ry * ry + rz * rz;
#ifdef BTBA_EXACT_DIV
        const float rho = m * ((e2 <= delta2) ? 1.0f : D.robust_delta / sqrtf(e2));
#else
        const float rho = m * ((e2 <= delta2) ? 1.0f : D.robust_delta * __builtin_amdgcn_rsqf(e2));
#endif
        acc[0] += m;
        acc[1] += wix; acc[2] += wiy; acc[3] += wiz;
        acc[4] += wjx; acc[5] += wjy; acc[6] += wjz;
        acc[7] += wix * wix; acc[8] += wix * wiy; acc[9] += wix * wiz; acc[10] += wiy * wiy; acc[11] += wiy * wiz; acc[12] += wiz * wiz;
        acc[13] += wjx * wjx; acc[14] += wjx * wjy; acc[15] += wjx * wjz; acc[16] += wjy * wjy; acc[17] += wjy * wjz; acc[18] += wjz * wjz;
        acc[19] += wix * wjx; acc[20] += wix * wjy; acc[21] += wix * wjz;
        acc[22] += wiy * wjx; acc[23] += wiy * wjy; acc[24] += wiy * wjz;
        acc[25] += wiz * wjx; acc[26] += wiz * wjy; acc[27] += wiz * wjz;
        acc[28] += rho * rx; acc[29] += rho * ry; acc[30] += rho * rz;
        acc[31] += rho * (wiy * rz - wiz * ry); acc[32] += rho * (wiz * rx - wix * rz); acc[33] += rho * (wix * ry - wiy * rx);
        acc[34] += rho * (wjy * rz - wjz * ry); acc[35] += rho * (wjz * rx - wjx * rz); acc[36] += rho * (wjx * ry - wjy * rx);
        acc[37] += rho;
        acc[38] += rho * (wiy * wiy + wiz * wiz); acc[39] += rho * (wix * wix + wiz * wiz); acc[40] += rho * (wix * wix + wiy * wiy);
        acc[41] += rho * (wjy * wjy + wjz * wjz); acc[42] += rho * (wjx * wjx + wjz * wjz); acc[43] += rho * (wjx * wjx + wjy * wjy);
    };
    // two correspondences per lane per trip (entries e and e + 256: every load instruction of a wave covers consecutive entries), all of
    // their loads in flight together
    if (C24) {
        const float2 *cb = reinterpret_cast<const float2 *>(corr);
        const size_t e_base = (size_t)D.corr_entry0 + (size_t)b * (size_t)D.corr_stride;
        // (three and four entries per lane per trip -- more bytes in flight -- measured slower everywhere: the registers they hold push the
        // fused kernel into spills, gpurun_out/r03_16)
        for (uint32_t e = lo + tid; e < hi; e += 2 * kBlock) {
            const uint32_t e2 = e + kBlock;
            const bool live2 = e2 < hi;
            const float2 *qa = cb + corr24_index(e_base + e), *qb = cb + corr24_index(e_base + (live2 ? e2 : e));
            const float2 a0 = ld_stream_f2<NT>(qa), a1 = ld_stream_f2<NT>(qa + 64), a2 = ld_stream_f2<NT>(qa + 128), b0 = ld_stream_f2<NT>(qb), b1 = ld_stream_f2<NT>(qb + 64), b2 = ld_stream_f2<NT>(qb + 128);
            accumulate(__float_as_uint(a0.x) != 0xFFFFFFFFu, a0.x, a0.y, a1.x, a1.y, a2.x, a2.y);
            accumulate(live2 && __float_as_uint(b0.x) != 0xFFFFFFFFu, b0.x, b0.y, b1.x, b1.y, b2.x, b2.y);
        }
    } else {
        const float4 *cb = corr + 2 * (size_t)b * D.corr_stride;
        bool misplaced = false;                       // an entry whose (imgIdx_i, imgIdx_j) is not this segment's pair
        float2 *const c24 = D.corr24_out;             // (wave-uniform) the re-layout of the first iteration: k_pack_corr24_batch's records, written as the entries pass by
        const size_t e_out = (size_t)D.corr_entry0 + (size_t)b * (size_t)D.corr_stride;
        auto entry = [&](const float4 &q0, const float4 &q1, bool live, uint32_t e) {
            // q0 = (imgIdx_i, imgIdx_j, pos_i.x, pos_i.y)  q1 = (pos_i.z, pos_j.x, pos_j.y, pos_j.z)
            const bool valid = live && __float_as_uint(q0.x) != 0xFFFFFFFFu;      // EntryJ::isValid
            misplaced |= valid & ((__float_as_uint(q0.x) != (uint32_t)fi) | (__float_as_uint(q0.y) != (uint32_t)fj));
            if (c24 && live) {
                float2 *o = c24 + corr24_index(e_out + e);
                o[0] = make_float2(valid ? q0.z : __uint_as_float(0xFFFFFFFFu), q0.w);
                o[64] = make_float2(q1.x, q1.y);
                o[128] = make_float2(q1.z, q1.w);
            }
            accumulate(valid, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w);
        };
        for (uint32_t e = lo + tid; e < hi; e += 2 * kBlock) {
            const uint32_t e2 = e + kBlock;
            const bool live2 = e2 < hi;
            const uint32_t e2c = live2 ? e2 : e;
            const float4 a0 = ld_stream_f4<NT>(cb + 2 * (size_t)e), a1 = ld_stream_f4<NT>(cb + 2 * (size_t)e + 1), b0 = ld_stream_f4<NT>(cb + 2 * (size_t)e2c), b1 = ld_stream_f4<NT>(cb + 2 * (size_t)e2c + 1);
            entry(a0, a1, true, e);
            entry(b0, b1, live2, e2c);
        }
        if (D.order_flag && misplaced) atomicOr(D.order_flag, 1);
    }
    float *out = partials + ((unsigned)b * D.sp_stride + (unsigned)(D.atomic_sums ? p : p * D.sparse_chunks + chunk) * (unsigned)kSparseVals);
    block_reduce_store<kSparseVals, 4>(acc, red, out, D.atomic_sums ? 1 : D.publish ? 2 : 0);
}
__device__ __forceinline__ void sparse_block(const SolveDims &D, const float4 *__restrict__ corr, const uint32_t *__restrict__ pair_offsets,
                                             const float *__restrict__ T, float *__restrict__ partials, int chunk, int p, int b, float *red)
{
    // (four instances of one loop; the choice is uniform over the workgroup)
    if (D.corr24) { if (b >= D.corr_nt_from) sparse_block_impl<true, true>(D, corr, pair_offsets, T, partials, chunk, p, b, red); else sparse_block_impl<true, false>(D, corr, pair_offsets, T, partials, chunk, p, b, red); }
    else { if (b >= D.corr_nt_from) sparse_block_impl<false, true>(D, corr, pair_offsets, T, partials, chunk, p, b, red); else sparse_block_impl<false, false>(D, corr, pair_offsets, T, partials, chunk, p, b, red); }
}

// EntryJ segments -> 24-byte correspondences, in place in the entry index space (entry e of the input is entry e of the output).
// grid (ceil(longest segment / 256), n_segments); seg[s] = (source offset, destination offset, length, i << 16 | j), offsets in entries.
// A valid entry that does not carry (i, j) raises the order flag (the array was not pair-major: the caller falls back to bucketing).
__global__ void __launch_bounds__(256) k_pack_corr24(const uint4 *__restrict__ seg, const uint4 *__restrict__ src, float2 *__restrict__ dst, int *__restrict__ order_flag)
{
    const uint4 d = seg[blockIdx.y];
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= d.z) return;
    const uint4 a = src[2 * (size_t)(d.x + e)], b = src[2 * (size_t)(d.x + e) + 1];
    const bool valid = a.x != 0xFFFFFFFFu;
    if (valid && ((a.x != (d.w >> 16)) | (a.y != (d.w & 0xFFFFu)))) atomicOr(order_flag, 1);
    float2 *o = dst + corr24_index((size_t)(d.y + e));
    o[0] = make_float2(valid ? __uint_as_float(a.z) : __uint_as_float(0xFFFFFFFFu), __uint_as_float(a.w));
    o[64] = make_float2(__uint_as_float(b.x), __uint_as_float(b.y));
    o[128] = make_float2(__uint_as_float(b.z), __uint_as_float(b.w));
}
// the same for a batch of pair-major instances described by their offset tables: grid (ceil(longest segment / 256), P, B)
__global__ void __launch_bounds__(256) k_pack_corr24_batch(int n_frames, int n_pairs, int64_t corr_stride, const uint32_t *__restrict__ pair_offsets,
                                                          const uint4 *__restrict__ src, float2 *__restrict__ dst, int *__restrict__ order_flag)
{
    const int p = blockIdx.y, b = blockIdx.z;
    const uint32_t *off = pair_offsets + (size_t)b * (n_pairs + 1);
    const uint32_t seg0 = off[p], len = off[p + 1] - seg0, e = blockIdx.x * 256u + threadIdx.x;
    if (e >= len) return;
    int fi, fj;
    pair_from_index(p, n_frames, fi, fj);
    const size_t at = (size_t)b * (size_t)corr_stride + seg0 + e;
    const uint4 a = src[2 * at], q = src[2 * at + 1];
    const bool valid = a.x != 0xFFFFFFFFu;
    if (order_flag && valid && ((a.x != (uint32_t)fi) | (a.y != (uint32_t)fj))) atomicOr(order_flag, 1);
    float2 *o = dst + corr24_index(at);
    o[0] = make_float2(valid ? __uint_as_float(a.z) : __uint_as_float(0xFFFFFFFFu), __uint_as_float(a.w));
    o[64] = make_float2(__uint_as_float(q.x), __uint_as_float(q.y));
    o[128] = make_float2(__uint_as_float(q.z), __uint_as_float(q.w));
}

// grid (sparse_chunks, P, B).  Workgroup (c, p, b) owns slice c of pair p's contiguous EntryJ segment.
__global__ void __launch_bounds__(kBlock) k_sparse_sweep(SolveDims D, const float4 *__restrict__ corr, const uint32_t *__restrict__ pair_offsets,
                                                        const float *__restrict__ T, float *__restrict__ partials)
{
    __shared__ float red[kRedFloats];
    sparse_block(D, corr, pair_offsets, T, partials, blockIdx.x, blockIdx.y, blockIdx.z, red);
}

// ---- dense sweep ------------------------------------------------------------------------------
// The reference is built with nvcc -use_fast_math (CMakeLists.txt:7): its divisions, sqrt and rsqrt are the
// approximate hardware forms.  v_rcp_f32 / v_rsq_f32 (1 ulp) are the CDNA counterparts; an IEEE division
// costs ~10 VALU instructions and this kernel had ~25 of them per pixel.
// -DBTBA_EXACT_DIV (an EXPERIMENT build, tests/tools/exact_div_experiment.py -- never the product): IEEE division and square root instead, to measure
// how much of the difference between this path's accept decisions and the oracle's the 1-ulp forms explain (profiles/r04/exact_div_experiment.json).
#ifdef BTBA_EXACT_DIV
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float fast_rsq(float x) { return 1.0f / sqrtf(x); }
#else
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
#endif

// XCD-aware remap of a 1-D grid: the dispatcher places block L on XCD L % 8 (observed, used for speed only),
// so logical work item L' = (contiguous range per XCD) keeps one instance's frames in ONE XCD's L2 instead of
// replicating them into all eight.
__device__ __forceinline__ unsigned xcd_remap(unsigned L, unsigned n)
{
    const unsigned q = n >> 3, r = n & 7u, xcd = L & 7u, slot = L >> 3;
    return xcd < r ? xcd * (q + 1) + slot : r * (q + 1) + (xcd - r) * q + slot;
}

// ---- per-pixel work of the dense sweep, branch-free so that all ten loads of a pixel are independent and
// in flight together.  The kernel is VALU-bound (PMC: VALU pipe ~87 % busy at 283 instructions per pixel),
// so the per-pixel arithmetic is kept minimal:
//   * the Jacobian row is accumulated in the TARGET CAMERA frame, a' = [-n_i ; n_i x q]; the model-frame row
//     of the reference is a = M a' with the constant per-pair 6x6  M = [[R_i, 0], [[t_i]x R_i, R_i]]
//     (n_w = R_i n_i, w = R_i q + t_i  =>  n_w x w = R_i (n_i x q) + [t_i]x R_i (-n_i)), so
//     S = M S' M^T and g = M g' are applied once per pair in k_system_solve instead of 18 FMAs per pixel;
//   * one set of bilinear tap coefficients serves camPos and normals (same taps, same in-image tests);
//   * the -inf sentinel tests of ICPUtil.h:96-102 are dropped: the cache never contains -inf (invalid data is
//     zeros and is blended in -- SURVEY.md appendix A.4 step 4; the reference notes the test never fires);
//   * the w lanes are not fetched: camPos.w is never used and normal.w is 0 by the wire format (Frame.h:75).
struct DenseCtx {
    Mat4 Tij;
    const float4 *cam_t, *nrm_t;
    float fx, fy, cx, cy, depth_min, depth_max, normal_thresh, dist2_thresh, delta, delta2, w_dense;
    int W, H;
};

struct PixelGeom {                  // stage 1: everything that does not need the target taps
    float qx, qy, qz, nqx, nqy, nqz;
    float c00, c10, c01, c11;       // final bilinear coefficients (row/column renormalisation folded in)
    int i00, i10, i01, i11;
    int xa, xb, ya, yb;             // clamped tap columns / rows (compact cache: needed to back-project the taps)
    bool valid;
};

__device__ __forceinline__ PixelGeom pixel_geom(const DenseCtx &C, const float4 &cs, const float4 &ns)
{
    PixelGeom g;
    g.valid = (cs.z > C.depth_min && cs.z < C.depth_max);
    const Mat4 &M = C.Tij;
    g.nqx = M.m[0] * ns.x + M.m[1] * ns.y + M.m[2] * ns.z;
    g.nqy = M.m[4] * ns.x + M.m[5] * ns.y + M.m[6] * ns.z;
    g.nqz = M.m[8] * ns.x + M.m[9] * ns.y + M.m[10] * ns.z;
    xform_point(M, cs.x, cs.y, cs.z, g.qx, g.qy, g.qz);
    const float rqz = fast_rcp(g.qz);
    float u = g.qx * C.fx * rqz + C.cx, v = g.qy * C.fy * rqz + C.cy;
    // in-image test of the rounded coordinates (SolverBundlingDenseUtil.h:91-94) without rounding them:
    // 0 <= (int)roundf(u) < W  <=>  -0.5 < u < W - 0.5  (round half away from zero; W - 0.5 is exact in fp32).
    // NaN / inf coordinates of rejected pixels fail the comparisons, so nothing non-finite reaches the int conversions.
    g.valid = g.valid && (u > -0.5f) && (u < (float)C.W - 0.5f) && (v > -0.5f) && (v < (float)C.H - 0.5f);
    u = g.valid ? u : 0.0f; v = g.valid ? v : 0.0f;
    // bilinear taps (ICPUtil.h:83-110): out-of-image taps get weight 0, weights renormalised per row, then per column
    const float fx0 = floorf(u), fy0 = floorf(v);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float alpha = u - fx0, beta = v - fy0;
    const bool okx0 = (unsigned)x0 < (unsigned)C.W, okx1 = (unsigned)(x0 + 1) < (unsigned)C.W;
    const bool oky0 = (unsigned)y0 < (unsigned)C.H, oky1 = (unsigned)(y0 + 1) < (unsigned)C.H;
    const int xa = min(max(x0, 0), C.W - 1), xb = min(max(x0 + 1, 0), C.W - 1);
    const int ya = min(max(y0, 0), C.H - 1), yb = min(max(y0 + 1, 0), C.H - 1);
    g.i00 = ya * C.W + xa; g.i10 = ya * C.W + xb; g.i01 = yb * C.W + xa; g.i11 = yb * C.W + xb;
    g.xa = xa; g.xb = xb; g.ya = ya; g.yb = yb;
    const float a0 = okx0 ? 1.0f - alpha : 0.0f, a1 = okx1 ? alpha : 0.0f;
    const float wr = a0 + a1;                                   // same for both rows
    const float b0 = (oky0 && wr > 0.0f) ? 1.0f - beta : 0.0f, b1 = (oky1 && wr > 0.0f) ? beta : 0.0f;
    const float ww = b0 + b1;
    const float k = (ww > 0.0f) ? fast_rcp(wr) * fast_rcp(ww) : 0.0f;
    const float k0 = k * b0, k1 = k * b1;
    g.c00 = k0 * a0; g.c10 = k0 * a1; g.c01 = k1 * a0; g.c11 = k1 * a1;
    return g;
}

__device__ __forceinline__ void pixel_accumulate(const DenseCtx &C, const PixelGeom &g,
                                                 const float4 &c00, const float4 &c10, const float4 &c01, const float4 &c11,
                                                 const float4 &n00, const float4 &n10, const float4 &n01, const float4 &n11,
                                                 float (&acc)[kDenseVals])
{
    const float cix = g.c00 * c00.x + g.c10 * c10.x + g.c01 * c01.x + g.c11 * c11.x;
    const float ciy = g.c00 * c00.y + g.c10 * c10.y + g.c01 * c01.y + g.c11 * c11.y;
    const float ciz = g.c00 * c00.z + g.c10 * c10.z + g.c01 * c01.z + g.c11 * c11.z;
    const float nix = g.c00 * n00.x + g.c10 * n10.x + g.c01 * n01.x + g.c11 * n11.x;
    const float niy = g.c00 * n00.y + g.c10 * n10.y + g.c01 * n01.y + g.c11 * n11.y;
    const float niz = g.c00 * n00.z + g.c10 * n10.z + g.c01 * n01.z + g.c11 * n11.z;
    const float dx = g.qx - cix, dy = g.qy - ciy, dz = g.qz - ciz;
    const float dist2 = dx * dx + dy * dy + dz * dz;
    const float dn = g.nqx * nix + g.nqy * niy + g.nqz * niz;
    // `&`, not `&&`: short-circuit evaluation turns into nested exec-mask branches with a block of zeroing moves on each
    const bool ok = g.valid & (ciz > C.depth_min) & (ciz < C.depth_max) & (dn >= C.normal_thresh) & (dist2 <= C.dist2_thresh);
    // rejected pixels contribute exact zeros.  Bit masks, not `ok ? x : 0`: the compiler turns a run of such selects into
    // an exec-mask branch with a block of zeroing moves on the other path; AND-ing with 0 / ~0 stays straight-line and is
    // NaN-safe (0 * NaN would poison the sums if a target normal were not finite; q is finite whenever the poses are)
    const unsigned keep = ok ? 0xFFFFFFFFu : 0u;
    auto masked = [keep](float x) { return __uint_as_float(__float_as_uint(x) & keep); };
    const float res = masked(-(dx * nix + dy * niy + dz * niz));
    const float e = res * res;
    const float wgt = masked(C.w_dense * ((e <= C.delta2) ? 1.0f : C.delta * fast_rsq(e)));
    // camera-frame row a' = [-n_i ; n_i x q]
    const float mx = masked(nix), my = masked(niy), mz = masked(niz);
    const float a[6] = { -mx, -my, -mz, my * g.qz - mz * g.qy, mz * g.qx - mx * g.qz, mx * g.qy - my * g.qx };
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const float wa = wgt * a[r];
#pragma unroll
        for (int c = r; c < 6; c++) acc[k++] += wa * a[c];
        acc[21 + r] += wa * res;
    }
    acc[27] += masked(1.0f);
}

#ifdef BTBA_REFERENCE_ORDER
// ---- TEST-ONLY build (-DBTBA_REFERENCE_ORDER, never the product; speed irrelevant): one pixel of the dense term evaluated in the REFERENCE's order --
// findDenseCorr (SolverBundlingDenseUtil.h:78-110): float4x4 * float3 / float4 as cuda_SimpleMatrixUtil.h:923-942 multiplies them, cameraToDepth with a
// division, the in-image test on the ROUNDED coordinates, bilinearInterpolationFloat4 (ICPUtil.h:83-110) with its weights renormalised per row and
// then per column, the distance as a square root against the threshold -- fused multiply-add contraction off, IEEE division and square root.  It is the
// oracle's find_dense_corr (oracle/btba_oracle.c) statement by statement.  What it is for: round 4's verdict asked whether the ORDER of the per-pixel
// arithmetic is what makes this path's accept decisions differ from the oracle's on identical inputs; with this pixel they must not differ at all
// (profiles/r05/reference_order_experiment.json).
__device__ __attribute__((noinline)) bool bilinear4_reference(float x, float y, const float4 *__restrict__ img, int W, int H, float out[4])
{
#pragma clang fp contract(off)
    const int p00x = (int)floorf(x), p00y = (int)floorf(y);
    const int p01x = p00x, p01y = p00y + 1, p10x = p00x + 1, p10y = p00y, p11x = p00x + 1, p11y = p00y + 1;
    const float alpha = x - (float)p00x, beta = y - (float)p00y;
    const float MINF = -INFINITY;
    float s0[4] = { 0.f, 0.f, 0.f, 0.f }, w0 = 0.0f;
    if ((unsigned)p00x < (unsigned)W && (unsigned)p00y < (unsigned)H) { const float4 q = img[(size_t)p00y * W + p00x]; const float v[4] = { q.x, q.y, q.z, q.w }; if (v[0] != MINF) { for (int k = 0; k < 4; k++) s0[k] += (1.0f - alpha) * v[k]; w0 += (1.0f - alpha); } }
    if ((unsigned)p10x < (unsigned)W && (unsigned)p10y < (unsigned)H) { const float4 q = img[(size_t)p10y * W + p10x]; const float v[4] = { q.x, q.y, q.z, q.w }; if (v[0] != MINF) { for (int k = 0; k < 4; k++) s0[k] += alpha * v[k]; w0 += alpha; } }
    float s1[4] = { 0.f, 0.f, 0.f, 0.f }, w1 = 0.0f;
    if ((unsigned)p01x < (unsigned)W && (unsigned)p01y < (unsigned)H) { const float4 q = img[(size_t)p01y * W + p01x]; const float v[4] = { q.x, q.y, q.z, q.w }; if (v[0] != MINF) { for (int k = 0; k < 4; k++) s1[k] += (1.0f - alpha) * v[k]; w1 += (1.0f - alpha); } }
    if ((unsigned)p11x < (unsigned)W && (unsigned)p11y < (unsigned)H) { const float4 q = img[(size_t)p11y * W + p11x]; const float v[4] = { q.x, q.y, q.z, q.w }; if (v[0] != MINF) { for (int k = 0; k < 4; k++) s1[k] += alpha * v[k]; w1 += alpha; } }
    float ss[4] = { 0.f, 0.f, 0.f, 0.f }, ww = 0.0f;
    if (w0 > 0.0f) { for (int k = 0; k < 4; k++) ss[k] += (1.0f - beta) * (s0[k] / w0); ww += (1.0f - beta); }
    if (w1 > 0.0f) { for (int k = 0; k < 4; k++) ss[k] += beta * (s1[k] / w1); ww += beta; }
    if (ww > 0.0f) { for (int k = 0; k < 4; k++) out[k] = ss[k] / ww; return true; }
    out[0] = out[1] = out[2] = out[3] = MINF;
    return false;
}

__device__ __attribute__((noinline)) void pixel_reference_order(const DenseCtx &C, float dist_thresh, const float4 &cs, const float4 &ns, float (&acc)[kDenseVals])
{
#pragma clang fp contract(off)
    const float *T = C.Tij.m;
    bool ok = (cs.z > C.depth_min && cs.z < C.depth_max) && (ns.x != -INFINITY);
    float q[3] = { 0.f, 0.f, 0.f }, ci[4] = { 0.f, 0.f, 0.f, 0.f }, ni[4] = { 0.f, 0.f, 0.f, 0.f };
    if (ok) {
        const float nt[4] = { T[0] * ns.x + T[1] * ns.y + T[2] * ns.z + T[3] * ns.w, T[4] * ns.x + T[5] * ns.y + T[6] * ns.z + T[7] * ns.w,
                              T[8] * ns.x + T[9] * ns.y + T[10] * ns.z + T[11] * ns.w, T[12] * ns.x + T[13] * ns.y + T[14] * ns.z + T[15] * ns.w };
        q[0] = T[0] * cs.x + T[1] * cs.y + T[2] * cs.z + T[3] * 1.0f;
        q[1] = T[4] * cs.x + T[5] * cs.y + T[6] * cs.z + T[7] * 1.0f;
        q[2] = T[8] * cs.x + T[9] * cs.y + T[10] * cs.z + T[11] * 1.0f;
        const float u = q[0] * C.fx / q[2] + C.cx, v = q[1] * C.fy / q[2] + C.cy;      // cameraToDepth, CUDACameraUtil.h:9-14
        const int sx = (int)roundf(u), sy = (int)roundf(v);
        ok = sx >= 0 && sy >= 0 && sx < C.W && sy < C.H;
        if (ok) {
            bilinear4_reference(u, v, C.cam_t, C.W, C.H, ci);
            ok = ci[2] > C.depth_min && ci[2] < C.depth_max;
        }
        if (ok) {
            bilinear4_reference(u, v, C.nrm_t, C.W, C.H, ni);
            ok = ni[0] != -INFINITY;
        }
        if (ok) {
            const float d[3] = { q[0] - ci[0], q[1] - ci[1], q[2] - ci[2] };
            const float dist = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const float dNormal = nt[0] * ni[0] + nt[1] * ni[1] + nt[2] * ni[2] + nt[3] * ni[3];
            ok = dNormal >= C.normal_thresh && dist <= dist_thresh;
        }
    }
    if (!ok) return;
    // the accepted pixel's contribution: the product's row and residual (pixel_accumulate) from these points
    const float dx = q[0] - ci[0], dy = q[1] - ci[1], dz = q[2] - ci[2];
    const float res = -(dx * ni[0] + dy * ni[1] + dz * ni[2]);
    const float e = res * res;
    const float wgt = C.w_dense * ((e <= C.delta2) ? 1.0f : C.delta / sqrtf(e));
    const float a[6] = { -ni[0], -ni[1], -ni[2], ni[1] * q[2] - ni[2] * q[1], ni[2] * q[0] - ni[0] * q[2], ni[0] * q[1] - ni[1] * q[0] };
    int k = 0;
    for (int r = 0; r < 6; r++) {
        const float wa = wgt * a[r];
        for (int c = r; c < 6; c++) acc[k++] += wa * a[c];
        acc[21 + r] += wa * res;
    }
    acc[27] += 1.0f;
}
#endif

// index of (r, c), r <= c, in the 21-entry upper-triangle row-major packing of a symmetric 6x6
__device__ __forceinline__ int tri21(int r, int c) { if (r > c) { const int t = r; r = c; c = t; } return r * 6 - r * (r - 1) / 2 + (c - r); }

// Epilogue of a dense workgroup: fixed-order reduction of the 28 per-lane sums over the workgroup, then the camera-frame ->
// model-frame congruence S = M S' M^T, g = M g' with M = [[R_i, 0], [[t_i]x R_i, R_i]] of the TARGET frame's pose of this iterate
// (see pixel_accumulate), then one 112-byte record per workgroup.  The congruence is linear, so applying it per tile and summing
// the tiles in k_system_solve equals applying it to the sum; doing it here (27 lanes, ~40 FMAs each, once per workgroup) takes it
// off the single-workgroup critical path of k_system_solve (it was 8.4 k of its 55 k cycles).
// M of the congruence (6 x 6, from the target frame's pose), staged in LDS by 36 threads at the START of a dense workgroup: its global
// loads are then off the tail of the workgroup (the epilogue used to wait a memory round trip for them)
__device__ __forceinline__ void dense_stage_M(float *red, const float *__restrict__ T_target)
{
    float *Mt = red + 4 * kDenseVals + kDenseVals + 4;
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 36) {
        const int e = (int)threadIdx.x - 64, r = e / 6, c = e % 6;
        float v;
        if (r < 3) v = (c < 3) ? T_target[4 * r + c] : 0.0f;
        else {
            const int q = r - 3, qa = (q + 1) % 3, qb = (q + 2) % 3;     // ([t]x R)[q][c] = t[qa] R[qb][c] - t[qb] R[qa][c]
            v = (c < 3) ? T_target[4 * qa + 3] * T_target[4 * qb + c] - T_target[4 * qb + 3] * T_target[4 * qa + c] : T_target[4 * q + (c - 3)];
        }
        Mt[e] = v;
    }
}

// FLIPPED: the sums were accumulated with a+ = D a, res+ = -res (dense_block_pinhole): S = D S+ D negates the nine entries that couple a
// translation row with a rotation row, g = -D g+ negates the three rotation entries -- exact.
// mode: 0 store the record, 1 add it to the record with float atomics (BTBA_REDUCE_ATOMIC), 2 store it WRITE-THROUGH at agent scope (k_chain:
// the record is read by another workgroup of the same launch, MI355X_MICROARCH.md "inter-workgroup visibility")
__device__ __forceinline__ void store_sum(float *p, float v, int mode)
{
    if (mode == 1) unsafeAtomicAdd(p, v);
    else if (mode == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool FLIPPED = false>
__device__ __forceinline__ void dense_epilogue(float (&acc)[kDenseVals], float *red, float *out, int mode = 0)
{
    const unsigned tid = item_tid();
    wave_fold_store<kDenseVals>(acc, red + (tid >> 6) * kDenseVals);
    __syncthreads();
    float *Sp = red + 4 * kDenseVals;                    // the workgroup's camera-frame sums (28) ...
    float *Mt = Sp + kDenseVals + 4;                     // ... and M (36)
    if (tid < kDenseVals) {
        float s = red[tid];
#pragma unroll
        for (int w = 1; w < 4; w++) s += red[w * kDenseVals + tid];
        if (FLIPPED) {
            // upper-triangle packing (tri21): row r < 3 holds columns r .. 5 from r * 6 - r (r - 1) / 2 on; its columns 3 .. 5 are the mixed entries
            const bool mixed = (tid >= 3 && tid < 6) || (tid >= 8 && tid < 11) || (tid >= 12 && tid < 15);
            if (mixed || (tid >= 24 && tid < 27)) s = -s;
        }
        Sp[tid] = s;
    }
    __syncthreads();
    const int idx = (int)tid;
    if (idx < 21) {
        int r = 0, rem = idx;
        while (rem >= 6 - r) { rem -= 6 - r; r++; }
        const int c = r + rem;
        float S[21];
#pragma unroll
        for (int k2 = 0; k2 < 21; k2++) S[k2] = Sp[k2];
        float Mr[6], Mc[6];
#pragma unroll
        for (int l = 0; l < 6; l++) { Mr[l] = Mt[6 * r + l]; Mc[l] = Mt[6 * c + l]; }
        float a = 0.0f;
#pragma unroll
        for (int k2 = 0; k2 < 6; k2++) {
            float u = 0.0f;                                   // u = (S' Mc^T)[k2]
#pragma unroll
            for (int l = 0; l < 6; l++) u += S[tri21(k2, l)] * Mc[l];
            a += Mr[k2] * u;
        }
        store_sum(out + idx, a, mode);
    } else if (idx < 27) {
        const int r = idx - 21;
        float a = 0.0f;
#pragma unroll
        for (int k2 = 0; k2 < 6; k2++) a += Mt[6 * r + k2] * Sp[21 + k2];
        store_sum(out + idx, a, mode);
    } else if (idx == 27) {
        store_sum(out + 27, Sp[27], mode);
    }
}

// 1-D grid of dense_tiles * Pd * B workgroups (XCD-remapped).  Lane = consecutive source pixel (coalesced
// float4 loads of the source camPos / normal); two pixels per lane per trip, sixteen target-tap gathers in
// flight; the taps stay in L1/L2 because neighbouring source pixels project to neighbouring target pixels.
__device__ __forceinline__ void dense_block(const SolveDims &D, const float4 *__restrict__ campos, const float4 *__restrict__ normals,
                                            const int2 ij, const float *__restrict__ T, const float *__restrict__ Tinv,
                                            float *__restrict__ partials, int tile, int p, int b, float *red)
{
    const int fi = ij.x, fj = ij.y;                       // fi = target, fj = source
    const size_t fb = (size_t)b * D.n_frames;
    const unsigned pb = (unsigned)b * (unsigned)D.pose_stride;
    dense_stage_M(red, T + pb + 16 * fi);
    DenseCtx C;
    C.Tij = mat_mul(load_mat4(Tinv + pb + 16 * fi), load_mat4(T + pb + 16 * fj));      // source camera -> target camera
    C.cam_t = campos + (fb + fi) * (size_t)D.npix; C.nrm_t = normals + (fb + fi) * (size_t)D.npix;
    C.fx = D.fx; C.fy = D.fy; C.cx = D.cx; C.cy = D.cy; C.depth_min = D.depth_min; C.depth_max = D.depth_max;
    C.normal_thresh = D.normal_thresh; C.dist2_thresh = D.dist_thresh * D.dist_thresh;
    C.delta = D.robust_delta; C.delta2 = D.robust_delta * D.robust_delta; C.w_dense = D.w_dense;
    C.W = D.width; C.H = D.height;
    const float4 *cam_s = campos + (fb + fj) * (size_t)D.npix, *nrm_s = normals + (fb + fj) * (size_t)D.npix;
    const int per = (D.npix + D.dense_tiles - 1) / D.dense_tiles;
    const int lo = min(D.npix, per * tile), hi = min(D.npix, per * (tile + 1));
    float acc[kDenseVals];
#pragma unroll
    for (int k = 0; k < kDenseVals; k++) acc[k] = 0.0f;

    {
        int s = lo + (int)threadIdx.x;
        float4 cs_n = make_float4(0.f, 0.f, 0.f, 0.f), ns_n = cs_n;
        if (s < hi) { cs_n = cam_s[s]; ns_n = nrm_s[s]; }
        for (; s < hi; s += kBlock) {
            const float4 cs = cs_n, ns = ns_n;
            if (s + kBlock < hi) { cs_n = cam_s[s + kBlock]; ns_n = nrm_s[s + kBlock]; }      // next pixel's stream loads
#ifdef BTBA_REFERENCE_ORDER
            pixel_reference_order(C, D.dist_thresh, cs, ns, acc);
            continue;
#endif
            const PixelGeom g = pixel_geom(C, cs, ns);
            // a wave whose 64 source pixels are all rejected (masked scenes: ~95 % of the image) skips the gathers
            if (__builtin_amdgcn_ballot_w64(g.valid) == 0ull) continue;
            const float4 c00 = C.cam_t[g.i00], c10 = C.cam_t[g.i10], c01 = C.cam_t[g.i01], c11 = C.cam_t[g.i11];
            const float4 n00 = C.nrm_t[g.i00], n10 = C.nrm_t[g.i10], n01 = C.nrm_t[g.i01], n11 = C.nrm_t[g.i11];
            pixel_accumulate(C, g, c00, c10, c01, c11, n00, n10, n01, n11, acc);
        }
    }
    float *out = partials + ((unsigned)b * D.dp_stride + (unsigned)(D.atomic_sums ? p : p * D.dense_tiles + tile) * (unsigned)kDenseVals);
    dense_epilogue(acc, red, out, D.atomic_sums ? 1 : D.publish ? 2 : 0);
}

// ---- per-block depth ranges of the compact cache ------------------------------------------------------------------
// [min, max] of the valid (gated: non-zero) depths per 8 x 8 pixel block of a cached frame; an empty block gets (+inf, -inf).
// Part of the frame cache: built once per frame (pool slot), not per solve.  One wave per block, lane = pixel, butterfly min / max;
// grid (blocks / 4, frames): one load per lane and no loop -- a single memory round trip deep.
__global__ void __launch_bounds__(kBlock) k_block_ranges(int width, int height, const float4 *__restrict__ zn, const int *__restrict__ slots, float2 *__restrict__ ranges)
{
    const int slot = slots ? slots[blockIdx.y] : (int)blockIdx.y;
    const int bw = width >> 3, nblk = bw * (height >> 3);
    const int lane = (int)threadIdx.x & 63, blk = (int)blockIdx.x * (kBlock / 64) + ((int)threadIdx.x >> 6);
    if (blk >= nblk) return;
    const int by = blk / bw, bx = blk - by * bw;
    const float d = zn[(size_t)slot * width * height + (size_t)((by * 8 + (lane >> 3)) * width + bx * 8 + (lane & 7))].x;
    const bool ok = d > 0.0f;                              // gated depth: 0 where invalid; NaN compares false
    float lo = ok ? d : INFINITY, hi = ok ? d : -INFINITY;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { lo = fminf(lo, __shfl_xor(lo, m)); hi = fmaxf(hi, __shfl_xor(hi, m)); }
    if (lane == 0) ranges[(size_t)slot * nblk + blk] = make_float2(lo, hi);
}

// Ordered list of the pixels of every frame that carry a depth (>= 0.1 m, the cache builder's validity rule).
// grid (B * K) x 1024.  The dense sweep walks this list instead of all Wd x Hd source pixels: a tracker's frames
// are masked to the object (~5 % of the image, src/Frame.cpp:342-358), so 95 % of the source stream disappears.
// Deterministic order (ascending pixel index).
constexpr int kListTrips = 32;        // ballots a wave keeps in scalar registers: frames up to 16 * 64 * 32 = 32 768 cached pixels
__global__ void __launch_bounds__(1024) k_valid_lists(int npix, const float4 *__restrict__ zn, uint32_t *__restrict__ lists, int *__restrict__ counts,
                                                     const int *__restrict__ slots)
{
    // one workgroup of 16 waves per frame; wave w owns the contiguous segment [w seg, (w+1) seg) and reads it in
    // coalesced 64-pixel trips, all loads in flight at once; the per-trip ballots stay in SGPRs, so after the 16
    // wave totals have been exchanged through LDS the lists are written without reading the frame again.
    __shared__ int wave_tot[16];
    const int f = slots ? slots[blockIdx.x] : (int)blockIdx.x;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const float4 *z = zn + (size_t)f * npix;
    uint32_t *out = lists + (size_t)f * npix;
    const int seg = (((npix + 15) / 16) + 63) & ~63, trips = seg / 64;
    const int s_wave = wave * seg;
    int total = 0;
    if (trips <= kListTrips) {
        unsigned long long m[kListTrips];
#pragma unroll
        for (int k = 0; k < kListTrips; k++) {
            const int s = s_wave + k * 64 + lane;
            const bool v = (k < trips) && (s < npix) && (z[min(s, npix - 1)].x >= 0.1f);      // (double)d >= 0.1  <=>  d >= 0.1f
            m[k] = __builtin_amdgcn_ballot_w64(v);
        }
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < kListTrips; k++) cnt += __popcll(m[k]);
        if (lane == 0) wave_tot[wave] = cnt;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { const int t = wave_tot[w]; if (w < wave) base += t; total += t; }
#pragma unroll
        for (int k = 0; k < kListTrips; k++) {
            const unsigned long long b = m[k];
            const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0));
            if ((b >> lane) & 1ull) out[base + before] = (uint32_t)(s_wave + k * 64 + lane);
            base += __popcll(b);
        }
    } else {                          // large caches: same scheme, the frame is read twice
        int cnt = 0;
        for (int k = 0; k < trips; k++) {
            const int s = s_wave + k * 64 + lane;
            cnt += __popcll(__builtin_amdgcn_ballot_w64((s < npix) && (z[min(s, npix - 1)].x >= 0.1f)));
        }
        if (lane == 0) wave_tot[wave] = cnt;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < 16; w++) { const int t = wave_tot[w]; if (w < wave) base += t; total += t; }
        for (int k = 0; k < trips; k++) {
            const int s = s_wave + k * 64 + lane;
            const bool v = (s < npix) && (z[min(s, npix - 1)].x >= 0.1f);
            const unsigned long long b = __builtin_amdgcn_ballot_w64(v);
            const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0));
            if (v) out[base + before] = (uint32_t)s;
            base += __popcll(b);
        }
    }
    if (threadIdx.x == 0) counts[f] = total;
}

// The same sweep on the compact cache: ONE 16-byte load per source pixel and per tap (5 loads instead of 10),
// camera-space points re-derived from z with the cache builder's exact arithmetic (zn_backproject).
template <bool SIMPLE, bool LISTS>
__device__ __forceinline__ void dense_block_zn(const SolveDims &D, const float4 *__restrict__ zn, const int2 ij,
                                               const float *__restrict__ T, const float *__restrict__ Tinv,
                                               float *__restrict__ partials, int tile, int p, int b, float *red,
                                               const uint32_t *__restrict__ valid_lists, const int *__restrict__ valid_counts, float *lut)
{
    // lut[0 .. Wd) = (float) full-res column of cache column x, lut[Wd .. Wd+Hd) the same for rows (SIMPLE: already
    // folded with the inverse intrinsics): the per-coordinate arithmetic becomes one LDS read
    for (int e = (int)threadIdx.x; e < D.width + D.height; e += kBlock) {
        const bool is_x = e < D.width;
        const float c = (float)(is_x ? zn_src_coord(e, D.zn_scale_w) : zn_src_coord(e - D.width, D.zn_scale_h));
        lut[e] = SIMPLE ? (is_x ? D.zn_ki[0] * c + D.zn_ki[2] : D.zn_ki[5] * c + D.zn_ki[6]) : c;      // see zn_backproject
    }
    __syncthreads();
    const float *lut_x = lut, *lut_y = lut + D.width;
    const int fi = ij.x, fj = ij.y;                       // fi = target, fj = source
    const size_t fb = (size_t)b * D.n_frames;
    const unsigned pb = (unsigned)b * (unsigned)D.pose_stride;
    dense_stage_M(red, T + pb + 16 * fi);
    DenseCtx C;
    C.Tij = mat_mul(load_mat4(Tinv + pb + 16 * fi), load_mat4(T + pb + 16 * fj));
    // the relative pose is the same in every lane: keep it in scalar registers (12 VGPRs back)
#pragma unroll
    for (int e = 0; e < 12; e++) C.Tij.m[e] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, C.Tij.m[e])));
    C.cam_t = nullptr; C.nrm_t = nullptr;
    C.fx = D.fx; C.fy = D.fy; C.cx = D.cx; C.cy = D.cy; C.depth_min = D.depth_min; C.depth_max = D.depth_max;
    C.normal_thresh = D.normal_thresh; C.dist2_thresh = D.dist_thresh * D.dist_thresh;
    C.delta = D.robust_delta; C.delta2 = D.robust_delta * D.robust_delta; C.w_dense = D.w_dense;
    C.W = D.width; C.H = D.height;
    // wave-uniform by construction; readfirstlane tells the compiler, so the frame bases live in SGPRs and the tap
    // gathers use the scalar-base + 32-bit-offset addressing form
    const size_t slot_t = (size_t)__builtin_amdgcn_readfirstlane((int)frame_slot_of(D, fb + fi));
    const size_t slot_s = (size_t)__builtin_amdgcn_readfirstlane((int)frame_slot_of(D, fb + fj));
    const float4 *zn_t = zn + slot_t * (size_t)D.npix, *zn_s = zn + slot_s * (size_t)D.npix;
    // LISTS: walk the source frame's ordered list of pixels that carry a depth (masked scenes: ~5 % of the image);
    // otherwise walk all pixels with incrementally advanced coordinates.  Two instantiations rather than one loop
    // with both: the kernel sits at the 96-VGPR / 5-waves-per-SIMD boundary and the merged loop measured 8 % slower.
    const int n_src = LISTS ? valid_counts[slot_s] : D.npix;
    const uint32_t *list = LISTS ? valid_lists + slot_s * (size_t)D.npix : nullptr;
    const int per = (n_src + D.dense_tiles - 1) / D.dense_tiles;
    const int lo = min(n_src, per * tile), hi = min(n_src, per * (tile + 1));
    const float inv_w = 1.0f / (float)D.width;
    float acc[kDenseVals];
#pragma unroll
    for (int k = 0; k < kDenseVals; k++) acc[k] = 0.0f;

    int t = lo + (int)threadIdx.x;
    int s_n = (t < hi) ? (LISTS ? (int)list[t] : t) : 0;
    int px_i = s_n % D.width, py_i = s_n / D.width;             // incremental coordinates (direct walk only)
    const int step_x = kBlock % D.width, step_y = kBlock / D.width;
    float4 zs_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < hi) zs_n = zn_s[s_n];
    for (; t < hi; t += kBlock) {
        const float4 zs = zs_n;
        const int s = s_n;
        if (t + kBlock < hi) { s_n = LISTS ? (int)list[t + kBlock] : t + kBlock; zs_n = zn_s[s_n]; }      // next pixel's stream loads
        int px, py;
        if (!LISTS) {
            px = px_i; py = py_i;
            px_i += step_x; py_i += step_y;
            if (px_i >= D.width) { px_i -= D.width; py_i++; }
        } else {
            py = (int)((float)s * inv_w);                       // s < 2^24: exact up to one unit, fixed up below
            py -= (py * D.width > s) ? 1 : 0;
            py += ((py + 1) * D.width <= s) ? 1 : 0;
            px = s - py * D.width;
        }
        const float3 cp = zn_backproject<SIMPLE>(D.zn_ki, lut_x[px], lut_y[py], zs.x);
        const PixelGeom g = pixel_geom(C, make_float4(cp.x, cp.y, cp.z, 1.0f), make_float4(zs.y, zs.z, zs.w, 0.0f));
        if (__builtin_amdgcn_ballot_w64(g.valid) == 0ull) continue;
        // 32-bit byte offsets from a scalar base: one shift per tap instead of a 64-bit shift-add (a frame is far below 4 GB)
        const char *zt = reinterpret_cast<const char *>(zn_t);
        const float4 z00 = *reinterpret_cast<const float4 *>(zt + ((unsigned)g.i00 << 4)), z10 = *reinterpret_cast<const float4 *>(zt + ((unsigned)g.i10 << 4));
        const float4 z01 = *reinterpret_cast<const float4 *>(zt + ((unsigned)g.i01 << 4)), z11 = *reinterpret_cast<const float4 *>(zt + ((unsigned)g.i11 << 4));
        const float xia = lut_x[g.xa], xib = lut_x[g.xb], yia = lut_y[g.ya], yib = lut_y[g.yb];
        const float3 c00 = zn_backproject<SIMPLE>(D.zn_ki, xia, yia, z00.x), c10 = zn_backproject<SIMPLE>(D.zn_ki, xib, yia, z10.x);
        const float3 c01 = zn_backproject<SIMPLE>(D.zn_ki, xia, yib, z01.x), c11 = zn_backproject<SIMPLE>(D.zn_ki, xib, yib, z11.x);
        pixel_accumulate(C, g, make_float4(c00.x, c00.y, c00.z, 1.f), make_float4(c10.x, c10.y, c10.z, 1.f), make_float4(c01.x, c01.y, c01.z, 1.f), make_float4(c11.x, c11.y, c11.z, 1.f),
                         make_float4(z00.y, z00.z, z00.w, 0.f), make_float4(z10.y, z10.z, z10.w, 0.f), make_float4(z01.y, z01.z, z01.w, 0.f), make_float4(z11.y, z11.z, z11.w, 0.f), acc);
    }
    float *out = partials + ((unsigned)b * D.dp_stride + (unsigned)(D.atomic_sums ? p : p * D.dense_tiles + tile) * (unsigned)kDenseVals);
    dense_epilogue(acc, red, out, D.atomic_sums ? 1 : D.publish ? 2 : 0);
}

// ---- the dense sweep for pinhole intrinsics on the GATED compact cache --------------------------------------------
// Same arithmetic per accepted pixel as dense_block_zn<true, .>, re-shaped around what the VALU of gfx950 actually charges
// (scripts/valu_calibrate.hip, profiles/r02/valu_calibration.md): a wave64 fp32 mul / add / fma with VGPR sources issues in
// 2 cycles, but v_cmp*, v_cndmask*, v_cvt*, v_floor, v_min / v_max, v_lshlrev, every VOP3 integer op (v_lshl_add_u32,
// v_mul_lo_u32, ...) and ANY VALU op with an SGPR source take 4, transcendentals 8.  The old loop spent 71 of its 222
// instructions on compares, selects and integer index arithmetic and 35 more on 2-cycle operations slowed down by an SGPR
// operand.  Here:
//   * the cache stores the GATED depth (0 where the reference's camPos is 0: d < 0.1 or NaN, CUDAImageUtil.cu:310-327),
//     so the per-tap `d >= 0.1 ? d : 0` select (5 compares + 5 selects per pixel) is done once, by the cache builder;
//   * a pinhole K^-1 has [15] == 1, so z = d without a multiply;
//   * the projection is CLAMPED to the image, uc = med3(u, 0, W - 1): the in-image test of the rounded coordinates
//     (-0.5 < u < W - 0.5, SolverBundlingDenseUtil.h:91-94) is |u - uc| < 0.5 -- exact, both differences are exact -- and
//     the bilinear taps are floor(uc), ceil(uc): a tap the reference skips because it lies outside the image
//     (ICPUtil.h:96-102) gets weight 0 here, and where the reference renormalises by the surviving weight, alpha / alpha,
//     the clamped coordinate has weight exactly 1.  No tap-validity compares, no weight selects, no reciprocals; when all
//     four taps are in the image the reference divides by (1 - alpha) + alpha = 1 +- 1 ulp, which is dropped (the
//     reference itself is built with -use_fast_math);
//   * tap addresses are computed in fp32 (exact: byte offsets < 2^24) and converted once per tap: 4 v_cvt instead of
//     8 clamps, 2 integer multiplies and 8 shift-adds; LDS look-up addresses likewise;
//   * depth-range tests are one unsigned compare of the bit patterns: for positive floats order is bit order, negative
//     values, zeros and NaN fall outside after the subtraction wraps;
//   * wave-uniform operands of 2-cycle operations (relative pose, intrinsics) live in VGPRs.
#ifdef BTBA_WG_TRACE
__shared__ unsigned long long wg_dbg[12];       // developer build: (end of prologue, end of pixel loop, live blocks) of the workgroup's dense item
#endif
struct PinholeCtx {
    float R[9], t[3];                 // relative pose source camera -> target camera (scalar registers)
    float fx, fy, cx, cy, wm1, hm1, wm2, hm2, w16, normal_thresh, dist2_thresh, wdelta, w_dense, ybase4;
    unsigned row16;                   // bytes per cache row
    unsigned zmin_bits, zrange_bits;  // depth_min < z < depth_max  <=>  bits(z) - (bits(depth_min) + 1) < zrange_bits (unsigned)
};

__device__ __forceinline__ float lds_f32_at(const float *base, unsigned byte_off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off); }
__device__ __forceinline__ float2 lds_f32x2_at(const float *base, unsigned byte_off) { const char *q = reinterpret_cast<const char *>(base) + byte_off; return make_float2(*reinterpret_cast<const float *>(q), *reinterpret_cast<const float *>(q + 4)); }
__device__ __forceinline__ float4 gather16(const float4 *base, unsigned byte_off) { return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + byte_off); }
// ... with the constant part of the address as the load's immediate offset (base + zext(byte_off) + IMM in 64 bits: a 32-bit `byte_off + IMM` may wrap, so the
// compiler cannot fold it and spends a vector add per tap)
template <int IMM> __device__ __forceinline__ float4 gather16_imm(const char *base, unsigned byte_off) { return *reinterpret_cast<const float4 *>(base + (size_t)byte_off + IMM); }
#ifndef BTBA_POSE_LDS
#define BTBA_POSE_LDS 0      // measured, round 6 (profiles/r06/sweep_diet.json): 16 fewer scalar-operand instructions per trip, 4 more broadcast LDS reads: 161 -> 169 us per launch.  Off.
#endif
#ifndef BTBA_POSE_VGPR
#define BTBA_POSE_VGPR 0
#endif
#ifndef BTBA_TAP_BASES
#define BTBA_TAP_BASES 1
#endif
#ifndef BTBA_STREAM_SADDR
#define BTBA_STREAM_SADDR 1
#endif
#ifndef BTBA_LUT_ABS
#define BTBA_LUT_ABS 1
#endif
#ifndef BTBA_LIST_LANES
#define BTBA_LIST_LANES 1
#endif
#ifndef BTBA_TAPS_EXEC
#define BTBA_TAPS_EXEC 1     // round 6 (profiles/r06/bound_probes.json, taps_exec.json): the four tap gathers only for lanes that passed the in-image / source-depth tests; same bits, -1.5 % on the launch
#endif
// LDS byte address of a pointer into the workgroup's LDS, and two consecutive floats at an absolute LDS byte address: the table base rides in the fp32 address
// arithmetic (exact below 2^24) instead of costing a vector add behind every float -> int conversion
__device__ __forceinline__ unsigned lds_address(const void *p) { return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char *)p; }
__device__ __forceinline__ float2 lds_f32x2_abs(unsigned addr) { const __attribute__((address_space(3))) float *q = (const __attribute__((address_space(3))) float *)(unsigned long)addr; return make_float2(q[0], q[1]); }
// opaque copy: keeps a value in a VGPR the compiler cannot see through
__device__ __forceinline__ unsigned opaque_vgpr(unsigned x) { asm volatile("" : "+v"(x)); return x; }
// v_min_f32 as the hardware does it: fminf() makes the compiler canonicalise its operands first (v_max_f32 x, x, x -- a half-rate
// instruction per operand, repeated in the loop even for loop invariants); the operands here are never signalling NaNs
__device__ __forceinline__ float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a * b + c as a v_fma_f32 whose destination is none of its sources.  An FMA that overwrites one of its own sources (v_fmac_f32, or v_fma_f32 with
// D = C) issues at HALF rate when its other two sources sit in VGPRs of the same parity or are the same register (profiles/r02/valu_calibration.md,
// rows `x:`); which registers they get is the allocator's business.  With the destination elsewhere the instruction is full rate whatever it does.
__device__ __forceinline__ float fma_nd(float a, float b, float c) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// ... with a wave-uniform bound held in a SCALAR register: v_min / v_med3 are half-rate whatever their operands, so the scalar operand costs
// no issue slot and the bound does not occupy a VGPR of a kernel that spills (left alone, the compiler keeps such loop invariants in VGPRs)
__device__ __forceinline__ float min_raw_s(float a, float bound) { float r; asm("v_min_f32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(bound)); return r; }
__device__ __forceinline__ float clamp0_s(float a, float bound) { float r; asm("v_med3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(bound)); return r; }     // med3(a, bound, 0); NaN -> 0
__device__ __forceinline__ float sgpr(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }

// column / row term of the back-projection of cache column / row e (e >= width: row e - width): K^-1[0][0] x_full(e) + K^-1[0][2] resp.
// K^-1[1][1] y_full + K^-1[1][2] -- the cache builder's nearest-neighbour source coordinate, zn_backproject<true>'s bracket
__device__ __forceinline__ float pinhole_lut_entry(const SolveDims &D, int e)
{
    const bool is_x = e < D.width;
    const float c = (float)(is_x ? zn_src_coord(e, D.zn_scale_w) : zn_src_coord(e - D.width, D.zn_scale_h));
    return is_x ? D.zn_ki[0] * c + D.zn_ki[2] : D.zn_ki[5] * c + D.zn_ki[6];
}

// Is the 8 x 8 block (bxl, byg) of the source frame PROVABLY without a pixel that lands in the target image?  A block's pixels with a
// usable depth lie in the frustum segment spanned by the block's extreme rays and its depth range [zlo, zhi] (k_block_ranges, clipped to
// depth_min .. depth_max); the relative pose maps the segment to a convex polytope whose vertices are the 8 transformed corners, and while
// every corner has q.z > 0 the projection (u, v) of any point inside lies within the corners' [min, max] in u and in v (a ratio of affine
// functions is quasi-linear on a convex set where the denominator is positive).  If that range misses the image by more than a margin --
// 0.01 pixel, three orders of magnitude above the rounding differences between this evaluation and the per-pixel one -- no pixel of the
// block can pass the in-image test of SolverBundlingDenseUtil.h:91-94 and the block contributes exactly nothing, as in the reference.
// Blocks without any usable depth go the same way.  (lut: the pinhole_lut_entry table; R, t: the relative pose.)
__device__ __forceinline__ bool block_is_live(const SolveDims &D, const float (&R)[9], const float (&t)[3], float2 zr, int bxl, int byg)
{
    zr.x = fmaxf(zr.x, D.depth_min); zr.y = fminf(zr.y, D.depth_max);      // usable depths: depth_min < z < depth_max
    bool live = zr.x <= zr.y;
    if (live) {
        // the block's extreme rays: column / row terms of its first and last column / row (the table entries, evaluated in place)
        const float xa = pinhole_lut_entry(D, 8 * bxl), xb = pinhole_lut_entry(D, 8 * bxl + 7);
        const float ya = pinhole_lut_entry(D, D.width + 8 * byg), yb = pinhole_lut_entry(D, D.width + 8 * byg + 7);
        float ulo = INFINITY, uhi = -INFINITY, vlo = INFINITY, vhi = -INFINITY, zq = INFINITY;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float x = (k & 1) ? xb : xa, y = (k & 2) ? yb : ya;
            const float rx = R[0] * x + R[1] * y + R[2], ry = R[3] * x + R[4] * y + R[5], rz = R[6] * x + R[7] * y + R[8];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float d = e ? zr.y : zr.x;
                const float qx = rx * d + t[0], qy = ry * d + t[1], qz = rz * d + t[2];
                const float rq = fast_rcp(qz);
                const float u = qx * D.fx * rq + D.cx, v = qy * D.fy * rq + D.cy;
                ulo = fminf(ulo, u); uhi = fmaxf(uhi, u); vlo = fminf(vlo, v); vhi = fmaxf(vhi, v); zq = fminf(zq, qz);
            }
        }
        const float m = 0.01f;
        const bool outside = (uhi < -0.5f - m) | (ulo > (float)D.width - 0.5f + m) | (vhi < -0.5f - m) | (vlo > (float)D.height - 0.5f + m);
        live = !(zq > 1e-3f && outside);          // NaN anywhere: comparisons false, the block stays
    }
    return live;
}

// The same test from the workgroup's ROTATED-RAY tables (colA[x] = R[:, 0] lx(x), rowB[y] = R[:, 1] ly(y) + R[:, 2], filled for the pixel loop anyway) and
// without a division.  A point q = d r + t of the block's hull, q.z > 0, projects left of the image by more than the margin m iff
//     fx q.x / q.z + cx < -0.5 - m   <=>   fx q.x + (cx + 0.5 + m) q.z < 0   <=>   d alpha_L(r) + beta_L < 0,   alpha_L = fx r.x + c_L r.z,  beta_L = fx t.x + c_L t.z,
// a LINEAR form in d for a fixed ray, so over the block's depth range it peaks at one of the two ends; likewise for the other three image edges.  The
// block is outside iff one of the four forms keeps its sign over all 4 corner rays x 2 depths: per ray 3 adds, 2 + 4 multiply-adds for the alphas, 8 products
// and 8 min / max instead of 2 x (9 multiply-adds, a reciprocal, 4 operations for (u, v) and 5 min / max) -- and no table entries recomputed per lane.
// 220 -> ~120 instructions of a workgroup's ~400-instruction prologue.  Same margin (0.01 pixel), same verdicts up to rounding far inside it.
__device__ __forceinline__ bool block_is_live_planes(const SolveDims &D, const float4 *colA, const float4 *rowB, const float (&t)[3], float2 zr, int bxl, int byg)
{
    zr.x = fmaxf(zr.x, D.depth_min); zr.y = fminf(zr.y, D.depth_max);      // usable depths: depth_min < z < depth_max
    bool live = zr.x <= zr.y;
    if (live) {
        const float4 ca = colA[8 * bxl], cb = colA[8 * bxl + 7], ra = rowB[8 * byg], rb = rowB[8 * byg + 7];
        const float m = 0.01f;
        const float cL = D.cx + 0.5f + m, cR = D.cx - ((float)D.width - 0.5f + m), cT = D.cy + 0.5f + m, cB = D.cy - ((float)D.height - 0.5f + m);
        float maxL = -INFINITY, minR = INFINITY, maxT = -INFINITY, minB = INFINITY, zq = INFINITY;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 cx4 = (k & 1) ? cb : ca, ry4 = (k & 2) ? rb : ra;
            const float rx = cx4.x + ry4.x, ry = cx4.y + ry4.y, rz = cx4.z + ry4.z;
            const float fxrx = rx * D.fx, fyry = ry * D.fy;
            const float aL = fxrx + cL * rz, aR = fxrx + cR * rz, aT = fyry + cT * rz, aB = fyry + cB * rz;
            maxL = fmaxf(maxL, fmaxf(aL * zr.x, aL * zr.y)); minR = fminf(minR, fminf(aR * zr.x, aR * zr.y));
            maxT = fmaxf(maxT, fmaxf(aT * zr.x, aT * zr.y)); minB = fminf(minB, fminf(aB * zr.x, aB * zr.y));
            zq = fminf(zq, fminf(rz * zr.x, rz * zr.y));
        }
        const float bL = D.fx * t[0] + cL * t[2], bR = D.fx * t[0] + cR * t[2], bT = D.fy * t[1] + cT * t[2], bB = D.fy * t[1] + cB * t[2];
        const bool outside = (maxL + bL < 0.0f) | (minR + bR > 0.0f) | (maxT + bT < 0.0f) | (minB + bB > 0.0f);
        live = !((zq + t[2] > 1e-3f) && outside);         // a corner at or behind the camera plane: the block stays
    }
    return live;
}

// (target, source, dense pair) of work position q: computed where the work order is the closed form (the default pair policies), read from
// the work table otherwise.  q is wave-uniform: scalar arithmetic / a scalar load.
__device__ __forceinline__ void dense_work_item(const SolveDims &D, int q, int &fi, int &fj, int &p)
{
    if (D.work_formula) {
        int d = 1, rem = q;                        // pairs at frame distance d: (i, i + d), i = 0 .. N-1-d, distances ascending
        while (rem >= D.n_frames - d) { rem -= D.n_frames - d; d++; }
        p = pair_index(rem, rem + d, D.n_frames);  // the canonical list IS the dense pair list of these policies
        fi = D.work_formula == 1 ? rem : rem + d;
        fj = D.work_formula == 1 ? rem + d : rem;
    } else {
        const int4 w = ld_const_i4(D.dense_work + q);
        fi = w.x; fj = w.y; p = w.z;
    }
}

// ---- the dense sweep for pinhole intrinsics on the GATED compact cache --------------------------------------------
// (the arithmetic per accepted pixel is dense_block_zn<true, .>'s; the shape is what the VALU of gfx950 charges for, see above)
template <int WALK>     // 0: 64-pixel row strips   1: the source frame's valid-pixel list   2: 8 x 8 pixel blocks per wave (cache width and height multiples of 8)
__device__ __forceinline__ void dense_block_pinhole(const SolveDims &D, const float4 *__restrict__ zn, int q,
                                                    const float *__restrict__ T, const float *__restrict__ Tinv,
                                                    float *__restrict__ partials, int tile, int b, float *red,
                                                    const uint32_t *__restrict__ valid_lists, const int *__restrict__ valid_counts, float *lut)
{
    const unsigned tid = item_tid();
    // The prologue is ONE round of memory loads, every address a function of the workgroup's grid position: the item (computed, or one
    // scalar load), the two poses (scalar loads: the previous launch wrote them), the target pose for M, the band's block ranges.
    // (Round 2: work table -> poses / ranges -> hull test -> list, four dependent round trips, 7.2 of a workgroup's 39 us.)
    int fi, fj, p;                                        // fi = target, fj = source
    dense_work_item(D, q, fi, fj, p);
    const size_t fb = (size_t)b * D.n_frames;
    const size_t slot_t = (size_t)__builtin_amdgcn_readfirstlane((int)frame_slot_of(D, fb + fi));
    const size_t slot_s = (size_t)__builtin_amdgcn_readfirstlane((int)frame_slot_of(D, fb + fj));
    const unsigned pb = (unsigned)b * (unsigned)D.pose_stride;
    const Mat4 Tinv_i = load_mat4_uniform(Tinv + pb + 16 * fi), T_j = load_mat4_uniform(T + pb + 16 * fj);
    float m_stage = 0.0f;                                 // M of the epilogue's congruence, from the target frame's pose (see dense_stage_M)
    if (tid >= 64 && tid < 64 + 36) {
        const float *T_target = T + pb + 16 * fi;
        const int e = (int)tid - 64, r = e / 6, c = e % 6;
        if (r < 3) m_stage = (c < 3) ? T_target[4 * r + c] : 0.0f;
        else {
            const int qq = r - 3, qa = (qq + 1) % 3, qb = (qq + 2) % 3;     // ([t]x R)[q][c] = t[qa] R[qb][c] - t[qb] R[qa][c]
            m_stage = (c < 3) ? T_target[4 * qa + 3] * T_target[4 * qb + c] - T_target[4 * qb + 3] * T_target[4 * qa + c] : T_target[4 * qq + (c - 3)];
        }
    }
    const int bw = D.width >> 3, bh = D.height >> 3;
    const int rows_per = (bh + D.dense_tiles - 1) / D.dense_tiles;
    const int r0 = min(bh, rows_per * tile), r1 = min(bh, rows_per * (tile + 1));
    const int nb = (WALK == 2) ? (r1 - r0) * bw : 0;      // blocks of this band (the host keeps it <= 4 x 256)
    const float2 *rng = (WALK == 2 && D.block_ranges) ? D.block_ranges + slot_s * (size_t)(bw * bh) + (size_t)r0 * bw : nullptr;
    const float2 zr0 = (rng && (int)tid < nb) ? rng[tid] : make_float2(0.0f, 0.0f);
    const float2 zr1 = (rng && (int)tid + kBlock < nb) ? rng[tid + kBlock] : make_float2(0.0f, 0.0f);      // a one-tile band of a 160 x 120 cache holds 300 blocks: both ranges of a lane in the SAME round of loads
                                                                                                           // (bands of more than 512 blocks fetch the rest in the compaction loop)
    const Mat4 Tij = mat_mul(Tinv_i, T_j);                // source camera -> target camera
    PinholeCtx C;
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) C.R[3 * r + c] = sgpr(Tij.m[4 * r + c]);
        C.t[r] = sgpr(Tij.m[4 * r + 3]);
    }
    // tables in LDS: lut[0 .. Wd) = K^-1[0][0] x_full(x) + K^-1[0][2] per cache column, lut[Wd .. Wd+Hd) the same for rows (the taps' back-projection),
    // and the ROTATED RAYS: the transformed point of source pixel (x, y) with depth d is q = d (R (lx, ly, 1)) + t = d (colA[x] + rowB[y]) + t with
    // colA[x] = R[:, 0] lx(x), rowB[y] = R[:, 1] ly(y) + R[:, 2] -- two 16-byte LDS reads, 3 adds and 3 FMAs per pixel instead of 2 reads,
    // 2 multiplies and 12 operations with a scalar-register operand (which issue at half rate, profiles/r02/valu_calibration.md).  One pass.
    float4 *colA = reinterpret_cast<float4 *>(lut + ((D.width + D.height + 3) & ~3));
    float4 *rowB = colA + D.width;
    for (int e = (int)tid; e < D.width + D.height; e += kBlock) {
        const float l = pinhole_lut_entry(D, e);
        lut[e] = l;
        if (e < D.width) colA[e] = make_float4(C.R[0] * l, C.R[3] * l, C.R[6] * l, 0.0f);
        else rowB[e - D.width] = make_float4(C.R[1] * l + C.R[2], C.R[4] * l + C.R[5], C.R[7] * l + C.R[8], 0.0f);
    }
    if (tid >= 64 && tid < 64 + 36) red[4 * kDenseVals + kDenseVals + 4 + ((int)tid - 64)] = m_stage;
    // Round 6: the pair's wave-uniform operands of the pixel loop -- relative pose and intrinsics -- ALSO in LDS, four 16-byte entries [R row | t] x 3,
    // (fx, fy, cx, cy).  A VALU instruction with a scalar-register source issues at HALF rate on gfx950 (profiles/r02/valu_calibration.md, rows `y:`), and
    // the loop has 16 of them per trip (3 translation FMAs, 4 projection multiply-adds, 9 for the normal rotation: 32 of a trip's 388 issue cycles); holding
    // the 16 values in VGPRs for the whole loop costs a wave per SIMD (archive 4.2: 101 VGPRs, slower).  Read back per trip as broadcast LDS loads
    // (LDS issue is not VALU issue) they live in VGPRs for a few instructions only.  Same operations on the same values: same bits.
    float4 *pose_l = rowB + D.height;
    if (tid == 0) {
        pose_l[0] = make_float4(C.R[0], C.R[1], C.R[2], C.t[0]); pose_l[1] = make_float4(C.R[3], C.R[4], C.R[5], C.t[1]);
        pose_l[2] = make_float4(C.R[6], C.R[7], C.R[8], C.t[2]); pose_l[3] = make_float4(D.fx, D.fy, D.cx, D.cy);
    }
    int *hdr = reinterpret_cast<int *>(pose_l + 4);                       // [0 .. 8) wave totals of the list compaction (two blocks per lane and pass: [0 .. 4) first halves, [4 .. 8) second)
    unsigned *blist = reinterpret_cast<unsigned *>(hdr + 8);
    int n_live = 0;
    if (WALK == 2) {
        // Blocks that are provably dead (block_is_live_planes: 90 % of the dead ones at c3) are removed BEFORE the walk; the live ones are compacted
        // into an ordered list in LDS (ballot + mbcnt, block order: deterministic) that the four waves share round-robin.
        __syncthreads();                                      // the ray tables are read by the test
        const int lane_c = (int)tid & 63, wave_c = __builtin_amdgcn_readfirstlane((int)tid >> 6);
        // Two blocks per lane and pass (idx, idx + 256): every band of a 160 x 120 cache -- 300 blocks at one tile -- is tested and compacted in ONE pass,
        // one exchange of wave totals, two barriers (round 4: two passes of 256, the second with a fresh round of range loads in front of it and two more
        // barriers: ~2 of a one-tile item's 9 us of set-up).  The list keeps its order -- blocks 0 .. 255 of the pass, then 256 .. 511 -- so the walk
        // and every sum are what they were.
        for (int c0 = 0; c0 < nb; c0 += 2 * kBlock) {
            const int idx0 = c0 + (int)tid, idx1 = idx0 + kBlock;
            bool live0 = false, live1 = false;
            unsigned code0 = 0, code1 = 0;
            if (idx0 < nb) {
                const int byl = idx0 / bw, bxl = idx0 - byl * bw, byg = r0 + byl;
                code0 = ((unsigned)byg << 16) | (unsigned)bxl;
                live0 = rng ? block_is_live_planes(D, colA, rowB, C.t, c0 ? rng[idx0] : zr0, bxl, byg) : true;
            }
            if (idx1 < nb) {
                const int byl = idx1 / bw, bxl = idx1 - byl * bw, byg = r0 + byl;
                code1 = ((unsigned)byg << 16) | (unsigned)bxl;
                live1 = rng ? block_is_live_planes(D, colA, rowB, C.t, c0 ? rng[idx1] : zr1, bxl, byg) : true;
            }
            const unsigned long long bal0 = __builtin_amdgcn_ballot_w64(live0), bal1 = __builtin_amdgcn_ballot_w64(live1);
            if (lane_c == 0) { hdr[wave_c] = __popcll(bal0); hdr[4 + wave_c] = __popcll(bal1); }
            __syncthreads();
            int base0 = n_live, total0 = 0, base1 = 0, total1 = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; w++) {
                const int t0 = hdr[w], t1 = hdr[4 + w];
                if (w < wave_c) { base0 += t0; base1 += t1; }
                total0 += t0; total1 += t1;
            }
            base1 += n_live + total0;
            const int before0 = __builtin_amdgcn_mbcnt_hi((unsigned)(bal0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal0, 0));
            const int before1 = __builtin_amdgcn_mbcnt_hi((unsigned)(bal1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal1, 0));
            if (live0) blist[base0 + before0] = code0;
            if (live1) blist[base1 + before1] = code1;
            n_live += total0 + total1;
            __syncthreads();
        }
    } else __syncthreads();
    C.fx = D.fx; C.fy = D.fy; C.cx = D.cx; C.cy = D.cy;
    // the clamp bounds come from the host: a float the kernel makes itself (v_cvt_f32_i32) lives in a VGPR for the whole loop; as kernel arguments
    // they are scalar operands of clamp0_s / min_raw_s
    C.wm1 = D.wm1; C.hm1 = D.hm1; C.wm2 = D.wm2; C.hm2 = D.hm2;
    C.w16 = 16.0f * (float)D.width; C.row16 = 16u * (unsigned)D.width;
    C.ybase4 = 4.0f * (float)D.width;                     // LDS byte offset of the row table
    C.normal_thresh = D.normal_thresh; C.dist2_thresh = D.dist2_thresh;
    C.wdelta = D.w_dense * D.robust_delta; C.w_dense = D.w_dense;
    C.zmin_bits = __float_as_uint(D.depth_min) + 1u; C.zrange_bits = __float_as_uint(D.depth_max) - __float_as_uint(D.depth_min) - 1u;
    const float4 *zn_t = zn + slot_t * (size_t)D.npix, *zn_s = zn + slot_s * (size_t)D.npix;
    constexpr bool LISTS = (WALK == 1);
    float acc[kDenseVals];
#pragma unroll
    for (int k = 0; k < kDenseVals; k++) acc[k] = 0.0f;

    // one source pixel: zs = its (gated depth, normal), ox / oy = LDS byte offsets of its column / row entries in the ray tables (from colA)
#ifdef BTBA_TRIP_TRACE
    unsigned long long tt_a = 0, tt_b = 0, tt_c = 0, tt_live = 0, tt_dead = 0;       // shader-clock sums of this wave: top of the trip / taps in flight / blend + accumulate
    auto stamp_after = [](float dep) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory"); return t; };
#endif
#ifdef BTBA_CENSUS
    unsigned cen[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };      // wave-uniform lane counts (scripts/sweep_census.py)
#endif
    typedef const volatile __attribute__((address_space(3))) btba_f4v lds_cv4;        // (an explicit LDS pointer: a volatile access through a generic one stays a flat_load)
    lds_cv4 *pose_v = (lds_cv4 *)pose_l;
#if BTBA_POSE_VGPR
    btba_f4v pv0 = (btba_f4v){ C.R[0], C.R[1], C.R[2], C.t[0] }, pv1 = (btba_f4v){ C.R[3], C.R[4], C.R[5], C.t[1] }, pv2 = (btba_f4v){ C.R[6], C.R[7], C.R[8], C.t[2] }, pvk = (btba_f4v){ D.fx, D.fy, D.cx, D.cy };
    asm volatile("" : "+v"(pv0), "+v"(pv1), "+v"(pv2), "+v"(pvk));
#endif
    const char *tap_row0 = reinterpret_cast<const char *>(zn_t), *tap_row1 = tap_row0 + C.row16;
    const float lut_addr_f = (float)lds_address(lut), ybase4_abs = C.ybase4 + lut_addr_f;
    auto pixel = [&](const float4 &zs, unsigned ox, unsigned oy) {
#ifdef BTBA_TRIP_TRACE
        unsigned long long tt0; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt0) :: "memory");
#endif
        // source pixel -> camera space (gated depth: 0 where invalid), depth-range test on the bit pattern
        const float d = zs.x;
        const bool src_ok = (__float_as_uint(d) - C.zmin_bits) < C.zrange_bits;
        // transform the point, project
        const float4 ra = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(colA) + ox), rb = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(colA) + oy);
#if BTBA_POSE_VGPR
        const btba_f4v P0 = pv0, P1 = pv1, P2 = pv2, PK = pvk;                              // developer experiment: the sixteen operands in VGPRs for the whole loop (needs 5 waves per SIMD: -DBTBA_FUSED_WAVES=5)
        const float qx = fma_nd(ra.x + rb.x, d, P0.w), qy = fma_nd(ra.y + rb.y, d, P1.w), qz = fma_nd(ra.z + rb.z, d, P2.w);
        const float rqz = fast_rcp(qz);
        const float u = __builtin_fmaf(qx * PK.x, rqz, PK.z), v = __builtin_fmaf(qy * PK.y, rqz, PK.w);      // NOT fma_nd: see below
#elif BTBA_POSE_LDS
        const btba_f4v P0 = pose_v[0], P1 = pose_v[1], P2 = pose_v[2], PK = pose_v[3];      // volatile: re-read every trip, never hoisted into loop-long registers
        // (fma_nd: with every operand in a VGPR the compiler overwrites the addend's register, and such a read-modify-write FMA issues at half rate when its
        // two multiplicands share a register parity -- the allocator's luck; a fresh destination is full rate whatever it gets)
        const float qx = fma_nd(ra.x + rb.x, d, P0.w), qy = fma_nd(ra.y + rb.y, d, P1.w), qz = fma_nd(ra.z + rb.z, d, P2.w);
        const float rqz = fast_rcp(qz);
        // (u, v through the compiler's own FMA, not fma_nd: an inline-asm consumer directly behind v_rcp_f32 is invisible to the hazard recognizer, which then
        // does not insert the wait state gfx950 needs between a transcendental and a VALU use of its result -- round 6's first build of this switch read a
        // stale reciprocal now and then: non-deterministic poses, 2e-4 off, caught by tests/test_cpp_bundler.py and scripts/r06/determinism.py)
        const float u = __builtin_fmaf(qx * PK.x, rqz, PK.z), v = __builtin_fmaf(qy * PK.y, rqz, PK.w);
#else
        const float qx = (ra.x + rb.x) * d + C.t[0], qy = (ra.y + rb.y) * d + C.t[1], qz = (ra.z + rb.z) * d + C.t[2];
        const float rqz = fast_rcp(qz);
        const float u = qx * C.fx * rqz + C.cx, v = qy * C.fy * rqz + C.cy;
#endif
        const float uc = clamp0_s(u, C.wm1), vc = clamp0_s(v, C.hm1);      // NaN -> 0: addresses stay in the frame
        const bool valid = src_ok & (fabsf(u - uc) < 0.5f) & (fabsf(v - vc) < 0.5f);
#ifdef BTBA_CENSUS
        cen[1] += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(src_ok)); cen[2] += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(valid));
        if (__builtin_amdgcn_ballot_w64(valid) == 0ull) cen[3] += 1u; else cen[4] += 64u;
#endif
#ifdef BTBA_TRIP_TRACE
        const unsigned long long tt1 = stamp_after(valid ? uc : vc);
        tt_a += tt1 - tt0;
        if (__builtin_amdgcn_ballot_w64(valid) == 0ull) { tt_dead++; return; }
#else
        if (__builtin_amdgcn_ballot_w64(valid) == 0ull) return;
#endif
        // rotate the normal (only the waves that go on need it: half of the block trips end above)
#if BTBA_POSE_LDS || BTBA_POSE_VGPR
        const float nqx = fma_nd(P0.z, zs.w, fma_nd(P0.y, zs.z, P0.x * zs.y));
        const float nqy = fma_nd(P1.z, zs.w, fma_nd(P1.y, zs.z, P1.x * zs.y));
        const float nqz = fma_nd(P2.z, zs.w, fma_nd(P2.y, zs.z, P2.x * zs.y));
#else
        const float nqx = C.R[0] * zs.y + C.R[1] * zs.z + C.R[2] * zs.w;
        const float nqy = C.R[3] * zs.y + C.R[4] * zs.z + C.R[5] * zs.w;
        const float nqz = C.R[6] * zs.y + C.R[7] * zs.z + C.R[8] * zs.w;
#endif
        // taps (x0, x0 + 1) x (y0, y0 + 1) with x0 = min(floor(uc), W - 2): at the right / bottom edge (uc = W - 1) the weights are (0, 1)
        // instead of (1, -) -- the same blend, and the four taps are always the 2 x 2 block at ONE computed address
        const float fx0 = min_raw_s(floorf(uc), C.wm2), fy0 = min_raw_s(floorf(vc), C.hm2);
        const float alpha = uc - fx0, beta = vc - fy0;
#if BTBA_TAP_BASES
        // one computed byte offset for the 2 x 2 block: the second row's base is a scalar add (wave-uniform), the + 16 the load's immediate offset field
        const unsigned o00 = (unsigned)(fy0 * C.w16 + 16.0f * fx0);                             // byte offset, fp32-exact below 2^24
#if BTBA_TAPS_EXEC
        // Lanes that already failed the in-image / source-depth tests (8.8 % in the trips that go on, clustered along block edges: profiles/r06/sweep_census.json) issue no
        // taps.  The texture addresser's service time per gather is exposed almost in full (profiles/r06/bound_probes.json: one more 16-byte gather per trip = +13.4 us
        // per launch, linear) and it skips idle quads.  The registers of those lanes stay undefined and are never used: `ok` contains `valid`, and `keep` zeroes
        // everything a rejected lane contributes (an AND: NaN-safe).
        float4 z00, z10, z01, z11;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wconditional-uninitialized"
        if (valid) { z00 = gather16_imm<0>(tap_row0, o00); z10 = gather16_imm<16>(tap_row0, o00); z01 = gather16_imm<0>(tap_row1, o00); z11 = gather16_imm<16>(tap_row1, o00); }
#pragma clang diagnostic pop
#else
        const float4 z00 = gather16_imm<0>(tap_row0, o00), z10 = gather16_imm<16>(tap_row0, o00), z01 = gather16_imm<0>(tap_row1, o00), z11 = gather16_imm<16>(tap_row1, o00);
#endif
#else
        const unsigned o00 = (unsigned)(fy0 * C.w16 + 16.0f * fx0), o01 = o00 + C.row16;      // byte offsets, fp32-exact below 2^24
        const float4 z00 = gather16(zn_t, o00), z10 = gather16(zn_t, o00 + 16u), z01 = gather16(zn_t, o01), z11 = gather16(zn_t, o01 + 16u);
#endif
#if BTBA_LUT_ABS
        const float2 xi2 = lds_f32x2_abs((unsigned)(4.0f * fx0 + lut_addr_f)), yi2 = lds_f32x2_abs((unsigned)(4.0f * fy0 + ybase4_abs));
#else
        const float2 xi2 = lds_f32x2_at(lut, (unsigned)(4.0f * fx0)), yi2 = lds_f32x2_at(lut, (unsigned)(4.0f * fy0 + C.ybase4));
#endif
        const float a0 = 1.0f - alpha, b0 = 1.0f - beta;
        // blend of the taps' camera-space points (x = column term * z, y = row term * z, z = gated depth) and normals; the
        // column / row terms are shared by the taps of a column / row, so they multiply the partial sums:
        //     cix = xi2.x (t00 + t01) + xi2.y (t10 + t11),  ciy = yi2.x (t00 + t10) + yi2.y (t01 + t11),  ciz = (t00 + t01) + (t10 + t11),  t = c z
        //     ni  = c00 z00.n + c10 z10.n + c01 z01.n + c11 z11.n
        // written out as the fused multiply-adds the compiler contracts these expressions to (x y + z w -> fma(x, y, z w), left to right --
        // the bits of every build since round 1), each as fma_nd: 13 of the 20 were v_fmac whose two multiplicands happened to sit in VGPRs
        // of the same parity, which issue at half rate (26 of a trip's ~400 issue cycles).
        const float c00 = b0 * a0, c10 = b0 * alpha, c01 = beta * a0, c11 = beta * alpha;
        const float t10 = c10 * z10.x, t01 = c01 * z01.x, t11 = c11 * z11.x;
        const float s0x = fma_nd(c00, z00.x, t01), s0y = fma_nd(c00, z00.x, t10);       // t00 + t01, t00 + t10
        const float s1x = fma_nd(c10, z10.x, t11), s1y = fma_nd(c01, z01.x, t11);       // t10 + t11, t01 + t11
        const float cix = fma_nd(xi2.x, s0x, xi2.y * s1x);
        const float ciy = fma_nd(yi2.x, s0y, yi2.y * s1y);
        const float ciz = s0x + s1x;
        const float nix = fma_nd(c11, z11.y, fma_nd(c01, z01.y, fma_nd(c00, z00.y, c10 * z10.y)));
        const float niy = fma_nd(c11, z11.z, fma_nd(c01, z01.z, fma_nd(c00, z00.z, c10 * z10.z)));
        const float niz = fma_nd(c11, z11.w, fma_nd(c01, z01.w, fma_nd(c00, z00.w, c10 * z10.w)));
#ifdef BTBA_TRIP_TRACE
        const unsigned long long tt3 = stamp_after(z00.x + z10.x + z01.x + z11.x);
        tt_b += tt3 - tt1;
#endif
        const float dx = qx - cix, dy = qy - ciy, dz = qz - ciz;
        const float dist2 = fma_nd(dz, dz, fma_nd(dx, dx, dy * dy));
        const float dn = fma_nd(nqz, niz, fma_nd(nqx, nix, nqy * niy));
        const bool ok = valid & ((__float_as_uint(ciz) - C.zmin_bits) < C.zrange_bits) & (dn >= C.normal_thresh) & (dist2 <= C.dist2_thresh);
#ifdef BTBA_CENSUS
        {
            const bool depth_ok = (__float_as_uint(ciz) - C.zmin_bits) < C.zrange_bits;
            cen[5] += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(ok)); cen[6] += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(valid & !depth_ok));
            cen[7] += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(valid & depth_ok & !ok));
        }
#endif
        // rejected pixels contribute exact zeros: AND with 0 / ~0 (one select, then 2-cycle v_and_b32; NaN-safe, unlike a multiply)
        const unsigned keep = opaque_vgpr(ok ? 0xFFFFFFFFu : 0u);
        auto masked = [keep](float x) { return __uint_as_float(__float_as_uint(x) & keep); };
        // SIGNS: the row is accumulated as a+ = [n_i ; n_i x q] and the residual as res+ = (q - c_i) . n_i, i.e. a = D a+ with
        // D = diag(-1, -1, -1, 1, 1, 1) and res = -res+ (SolverBundlingDenseUtil.h:78-110, LieDerivUtil.h:228-273).  Then S = D S+ D and
        // g = -D g+ : exact sign changes of whole sums, applied once per workgroup in the epilogue instead of four negations per pixel.
        const float res = masked(fma_nd(dz, niz, fma_nd(dx, nix, dy * niy)));
        // Huber (SolverBundlingUtil.h:24-40) times the dense weight: rho' = 1 for e <= delta^2, delta / sqrt(e) above  <=>  min(1, delta rsq(e)).
        // Not masked: a rejected pixel has res = 0, rsq(0) = inf, min(w, inf) = w -- finite, and it multiplies a masked (zero) row.
        const float wgt = min_raw_s(C.wdelta * fast_rsq(res * res), C.w_dense);
        const float mx = masked(nix), my = masked(niy), mz = masked(niz);
        const float a[6] = { mx, my, mz, my * qz - mz * qy, mz * qx - mx * qz, mx * qy - my * qx };
        int k = 0;
        // acc += wa_r * a_c is a read-modify-write FMA: it issues at full rate only when its two multiplicands sit in VGPRs of
        // different parity (profiles/r02/valu_calibration.md).  (wa_r, a_r) held as an aligned register PAIR puts every wa in an
        // even and every a in an odd register, whatever the allocator does with the rest.
        typedef float f2v __attribute__((ext_vector_type(2)));
        f2v pr[6];
#pragma unroll
        for (int r = 0; r < 6; r++) { pr[r] = (f2v){ wgt * a[r], a[r] }; asm volatile("" : "+v"(pr[r])); }
        f2v rr = (f2v){ wgt, res };
        asm volatile("" : "+v"(rr));
#pragma unroll
        for (int r = 0; r < 6; r++) {
#pragma unroll
            for (int c = r; c < 6; c++) acc[k++] += pr[r].x * pr[c].y;
            acc[21 + r] += pr[r].x * rr.y;
        }
        acc[27] += masked(1.0f);
#ifdef BTBA_TRIP_TRACE
        tt_c += stamp_after(acc[27] + acc[0] + acc[20] + acc[26]) - tt3; tt_live++;
#endif
    };

    if (WALK == 2) {
        // Waves walk 8 x 8 pixel blocks of this band of block rows; lane = (row, column) inside the block.  The region of a source
        // frame that projects into the target is a compact blob, so whole blocks fall outside it (51 % of them at c3; 64 x 1 strips:
        // 31 %), and a block's taps land in a compact patch of the target.  The list holds the blocks that are not provably dead.
        const int lane = (int)tid & 63, wave = __builtin_amdgcn_readfirstlane((int)tid >> 6);
        n_live = __builtin_amdgcn_readfirstlane(n_live);
        if (D.live_blocks && tid == 0) atomicAdd(D.live_blocks, (unsigned long long)n_live);
#ifdef BTBA_WG_TRACE
        if (tid == 0) { wg_dbg[0] = wall_clock64(); wg_dbg[2] = (unsigned long long)n_live; }
#endif
        const unsigned lx = (unsigned)lane & 7u, ly = (unsigned)lane >> 3;
        const unsigned lane_px = ly * (unsigned)D.width + lx;          // pixel offset of the lane inside its block
        const unsigned ox_l = 16u * lx, oy_l = 16u * ly + 16u * (unsigned)D.width;
        unsigned lane_off16 = 16u * lane_px;
        constexpr unsigned kBlockStep = 128u;            // 8 entries of 16 bytes
        // This wave's blocks: list entries wave, wave + 4, ... -- nt of them.  While block i is worked on, block i + 1's pixels are in flight.
        // The prefetch is UNCONDITIONAL (the last trip re-reads its own block): with a conditional one the number of loads in flight behind it
        // depended on the path taken and the compiler had to wait for ALL of them (`s_waitcnt vmcnt(0)` right behind the prefetch it had just
        // issued: the stream latency was paid in full on every trip, 2 % of the launch); now the wait for the current block's pixels is `vmcnt(1)`.
        // (Reading the list entry one more trip ahead, so that the prefetch is issued from a scalar register without an LDS round trip in
        // front of it: no change, r03 call 40.)
        // Two trips per loop iteration with the two register sets swapping roles: a single-trip loop rotates (next -> current) through
        // eight v_mov per trip, 5 % of its instructions.
        const int nt = (n_live > wave) ? (n_live - wave + kBlock / 64 - 1) / (kBlock / 64) : 0;
#if BTBA_LIST_LANES
        // this wave's list entries ride in a register, 64 at a time (lane l: entry 64 c + l of the wave), and a trip takes its entry with v_readlane_b32:
        // no LDS round trip (address move, ds_read, wait, readfirstlane) at the top of every trip
        int list_chunk = 0;
        unsigned list_reg = nt > 0 ? blist[wave + (kBlock / 64) * min(lane, nt - 1)] : 0u;
#endif
        auto fetch = [&](int i, unsigned &code, float4 &zs) {
#if BTBA_LIST_LANES
            const int idx = min(i, nt - 1);
            if ((idx >> 6) != list_chunk) { list_chunk = idx >> 6; list_reg = blist[wave + (kBlock / 64) * min(64 * list_chunk + lane, nt - 1)]; }
            code = (unsigned)__builtin_amdgcn_readlane((int)list_reg, idx & 63);
#else
            code = (unsigned)__builtin_amdgcn_readfirstlane((int)blist[wave + (kBlock / 64) * min(i, nt - 1)]);
#endif
#if BTBA_STREAM_SADDR
            // block base in the scalar address (wave-uniform), the lane's pixel inside the block as the constant vector offset: no vector add per trip
            // (the lane offset goes through an opaque copy: seen as loop-invariant, `frame base + lane offset` is hoisted as a 64-bit VECTOR address and the block
            // offset becomes a 64-bit vector add per trip)
            asm volatile("" : "+v"(lane_off16));          // (in place: no copy)
            zs = gather16_imm<0>(reinterpret_cast<const char *>(zn_s) + (size_t)(16u * ((code >> 16) * 8u * (unsigned)D.width + (code & 0xFFFFu) * 8u)), lane_off16);
#else
            zs = gather16(zn_s, 16u * ((code >> 16) * 8u * (unsigned)D.width + (code & 0xFFFFu) * 8u + lane_px));
#endif
        };
        auto work = [&](const float4 &zs, unsigned code) { pixel(zs, ox_l + kBlockStep * (code & 0xFFFFu), oy_l + kBlockStep * (code >> 16)); };
        if (nt > 0) {
            unsigned code_a, code_b = 0u;
            float4 zs_a, zs_b = make_float4(0.f, 0.f, 0.f, 0.f);
            fetch(0, code_a, zs_a);
            for (int i = 0;;) {
                fetch(i + 1, code_b, zs_b);
                work(zs_a, code_a);
                if (++i >= nt) break;
                fetch(i + 1, code_a, zs_a);
                work(zs_b, code_b);
                if (++i >= nt) break;
            }
        }
#ifdef BTBA_CENSUS
        if (D.live_blocks && lane == 0) for (int k = 1; k < 8; k++) atomicAdd(D.live_blocks + k, (unsigned long long)cen[k]);
#endif
#ifdef BTBA_WG_TRACE
        if (tid == 0) wg_dbg[1] = wall_clock64();
#endif
#ifdef BTBA_TRIP_TRACE
        if (tid == 0) { wg_dbg[4] = tt_a; wg_dbg[5] = tt_b; wg_dbg[6] = tt_c; wg_dbg[7] = tt_live | (tt_dead << 32); }
#endif
    } else {
        const int n_src = LISTS ? valid_counts[slot_s] : D.npix;
        const uint32_t *list = LISTS ? valid_lists + slot_s * (size_t)D.npix : nullptr;
        const int per = (n_src + D.dense_tiles - 1) / D.dense_tiles;
        const int lo = min(n_src, per * tile), hi = min(n_src, per * (tile + 1));
        const float inv_w = 1.0f / (float)D.width;
        int t = lo + (int)tid;
        int s_n = (t < hi) ? (LISTS ? (int)list[t] : t) : 0;
        // direct walk: LDS byte offsets of the pixel's column / row entries (row table behind the column table), advanced by
        // one workgroup stride per trip
        const unsigned w4 = 4u * (unsigned)D.width;
        unsigned px4 = 4u * (unsigned)(s_n % D.width), py4 = 4u * (unsigned)(s_n / D.width) + w4;
        const unsigned step_x4 = 4u * (unsigned)(kBlock % D.width), step_y4 = 4u * (unsigned)(kBlock / D.width);
        float4 zs_n = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < hi) zs_n = zn_s[s_n];
        for (; t < hi; t += kBlock) {
            const float4 zs = zs_n;
            const int s = s_n;
            // next pixel's stream loads, UNCONDITIONAL like the block walk's (the last trip re-reads the lane's last pixel): the wait for
            // this trip's taps then does not have to cover a load that only some paths issue
            const int tn = min(t + (int)kBlock, hi - 1);
            s_n = LISTS ? (int)list[tn] : tn; zs_n = zn_s[s_n];
            unsigned ox, oy;
            if (!LISTS) {
                ox = px4; oy = py4;
                px4 += step_x4; py4 += step_y4;
                if (px4 >= w4) { px4 -= w4; py4 += 4u; }
            } else {
                const float sf = (float)s;
                const float pyf = floorf((sf + 0.5f) * inv_w);              // exact: the margin 0.5 / W is far above the rounding of the product
                ox = (unsigned)(4.0f * (sf - pyf * (float)D.width)); oy = (unsigned)(4.0f * pyf + C.ybase4);
            }
            pixel(zs, 4u * ox, 4u * oy);                  // 4-byte table offsets -> 16-byte entries (rowB follows colA as the row table follows the column table)
        }
    }
    float *out = partials + ((unsigned)b * D.dp_stride + (unsigned)(D.atomic_sums ? p : p * D.dense_tiles + tile) * (unsigned)kDenseVals);
    dense_epilogue<true>(acc, red, out, D.atomic_sums ? 1 : D.publish ? 2 : 0);
}

template <bool SIMPLE, bool LISTS>
__global__ void __launch_bounds__(kBlock, 5) k_dense_sweep_zn(SolveDims D, const float4 *__restrict__ zn, const int2 *__restrict__ dense_pairs,
                                                             const float *__restrict__ T, const float *__restrict__ Tinv, float *__restrict__ partials,
                                                             const uint32_t *__restrict__ valid_lists, const int *__restrict__ valid_counts)
{
    __shared__ float red[kRedFloats];
    extern __shared__ __attribute__((aligned(16))) float zn_lut[];        // (Wd + Hd) floats
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = (int)(L % (unsigned)D.dense_tiles);
    const int p = (int)((L / (unsigned)D.dense_tiles) % (unsigned)D.n_dense_pairs);
    const int b = (int)(L / ((unsigned)D.dense_tiles * (unsigned)D.n_dense_pairs));
    // pinhole intrinsics: the same block as in the fused launch (same partial sums, bit for bit); `p` is then a WORK POSITION (dense_work_item)
    if (SIMPLE && LISTS) dense_block_pinhole<1>(D, zn, p, T, Tinv, partials, tile, b, red, valid_lists, valid_counts, zn_lut);
    else if (SIMPLE && D.walk_blocks) dense_block_pinhole<2>(D, zn, p, T, Tinv, partials, tile, b, red, valid_lists, valid_counts, zn_lut);
    else if (SIMPLE) dense_block_pinhole<0>(D, zn, p, T, Tinv, partials, tile, b, red, valid_lists, valid_counts, zn_lut);
    else dense_block_zn<false, LISTS>(D, zn, dense_pairs[p], T, Tinv, partials, tile, p, b, red, valid_lists, valid_counts, zn_lut);
}

// 1-D grid of dense_tiles * Pd * B workgroups (XCD-remapped).
__global__ void __launch_bounds__(kBlock, 3) k_dense_sweep(SolveDims D, const float4 *__restrict__ campos, const float4 *__restrict__ normals,
                                                              const int2 *__restrict__ dense_pairs, const float *__restrict__ T, const float *__restrict__ Tinv,
                                                              float *__restrict__ partials)
{
    __shared__ float red[kRedFloats];
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = (int)(L % (unsigned)D.dense_tiles);
    const int p = (int)((L / (unsigned)D.dense_tiles) % (unsigned)D.n_dense_pairs);
    const int b = (int)(L / ((unsigned)D.dense_tiles * (unsigned)D.n_dense_pairs));
    dense_block(D, campos, normals, dense_pairs[p], T, Tinv, partials, tile, p, b, red);
}

// Both sweeps of one Gauss-Newton iteration in ONE launch: n_d dense workgroups (VALU-bound) interleaved with
// n_s sparse workgroups (HBM-streaming) so the two overlap on every CU instead of running back to back.
// Per XCD x (blocks g = 8 s + x, s = slot): a contiguous range of dense items (L2 locality, as in xcd_remap)
// and every R_x-th slot a sparse item.
#ifndef BTBA_FUSED_WAVES
#define BTBA_FUSED_WAVES 3
#endif
template <int LAYOUT>   // 0: float4 camPos + float4 normals; compact cache: 1 zero-skew K, 2 general K, 3 / 4 the same walking valid-pixel lists
__device__ __forceinline__ void fused_item(const SolveDims &D, unsigned n_d, unsigned n_s, unsigned g,
                                           const float4 *__restrict__ campos, const float4 *__restrict__ normals,
                                           const int2 *__restrict__ dense_pairs, const float *__restrict__ T, const float *__restrict__ Tinv,
                                           float *__restrict__ dense_partials,
                                           const float4 *__restrict__ corr, const uint32_t *__restrict__ pair_offsets, float *__restrict__ sparse_partials,
                                           const uint32_t *__restrict__ valid_lists, const int *__restrict__ valid_counts, float *red, float *zn_lut)
{
    const unsigned G = n_d + n_s, xcd = g & 7u, slot = g >> 3;
    const unsigned qd = n_d >> 3, rd = n_d & 7u;
    const unsigned Gx = (G - xcd + 7u) >> 3, ndx = qd + (xcd < rd ? 1u : 0u), nsx = Gx - ndx;
    unsigned sparse_base = 0;                                   // sparse items owned by lower XCDs
    for (unsigned y = 0; y < xcd; y++) sparse_base += ((G - y + 7u) >> 3) - (qd + (y < rd ? 1u : 0u));
    // Sparse items: a share of them (D.sparse_tail_256 / 256) is HELD BACK to the end of the XCD's sequence, the rest is interleaved evenly
    // among the dense items.  A launch ends with its occupancy falling for ~30 us while the last long dense items finish (12-15 % of its
    // span); short, HBM-streaming sparse items fill those emptying slots, and the dense items start that much earlier.
    const unsigned nst = (nsx * (unsigned)D.sparse_tail_256) >> 8, nsi = nsx - nst, Gi = ndx + nsi;
    // (an even interleave period from 4 on is lowered to the odd one below it: with periods 4 and 6 the sparse items of an XCD kept landing on
    // the same compute units of the dispatcher's round robin -- 196 and 182 us per launch against 171 us at period 5, gpurun_out/r03_24;
    // period 2, the masked launch's plain alternation, measured fine)
    const unsigned Rx0 = nsi ? Gi / nsi : 0xFFFFFFFFu;
    const unsigned Rx = (nsi && Rx0 >= 4u && !(Rx0 & 1u)) ? Rx0 - 1u : Rx0;
    const bool in_tail = slot >= Gi;
    const bool is_sparse = in_tail || (nsi && (slot % Rx == 0u) && (slot / Rx < nsi));
    const unsigned sparse_local = in_tail ? nsi + (slot - Gi) : slot / Rx;
#ifdef BTBA_WG_TRACE
    const unsigned long long wg_t0 = wall_clock64();
#endif
    if (is_sparse) {
        const unsigned i = sparse_base + sparse_local;           // (chunk fastest, then pair, then instance)
        const int chunk = (int)(i % (unsigned)D.sparse_chunks);
        const int p = (int)((i / (unsigned)D.sparse_chunks) % (unsigned)D.n_pairs);
        const int b = (int)(i / ((unsigned)D.sparse_chunks * (unsigned)D.n_pairs));
        sparse_block(D, corr, pair_offsets, T, sparse_partials, chunk, p, b, red);
    } else {
        const unsigned before = nsi ? min(nsi, (slot + Rx - 1u) / Rx) : 0u;     // sparse slots of this XCD before `slot`
        const unsigned dl = slot - before;                                      // dense item local to the XCD
        const unsigned L = (xcd < rd ? xcd * (qd + 1u) : rd * (qd + 1u) + (xcd - rd) * qd) + dl;
        const int b = (int)(L / ((unsigned)D.dense_tiles * (unsigned)D.n_dense_pairs));
        const unsigned Lb = L - (unsigned)b * (unsigned)D.dense_tiles * (unsigned)D.n_dense_pairs;
        const int tile = D.tile_major ? (int)(Lb / (unsigned)D.n_dense_pairs) : (int)(Lb % (unsigned)D.dense_tiles);
        int p = D.tile_major ? (int)(Lb % (unsigned)D.n_dense_pairs) : (int)(Lb / (unsigned)D.dense_tiles);
        // work order: the pairs of a band heaviest first, so that the workgroups still running when the launch drains are short ones.
        // p is a WORK POSITION: the pinhole blocks compute their item from it (dense_work_item), the other layouts read the work table entry
        if (LAYOUT == 1 && D.walk_blocks) dense_block_pinhole<2>(D, campos, p, T, Tinv, dense_partials, tile, b, red, valid_lists, valid_counts, zn_lut);
        else if (LAYOUT == 1) dense_block_pinhole<0>(D, campos, p, T, Tinv, dense_partials, tile, b, red, valid_lists, valid_counts, zn_lut);     // `campos` carries the compact cache
        else if (LAYOUT == 3) dense_block_pinhole<1>(D, campos, p, T, Tinv, dense_partials, tile, b, red, valid_lists, valid_counts, zn_lut);
        else {
            const int4 w = D.dense_work[p];
            const int2 ij = make_int2(w.x, w.y);
            p = w.z;
            if (LAYOUT == 0) dense_block(D, campos, normals, ij, T, Tinv, dense_partials, tile, p, b, red);
            else if (LAYOUT == 2) dense_block_zn<false, false>(D, campos, ij, T, Tinv, dense_partials, tile, p, b, red, valid_lists, valid_counts, zn_lut);
            else dense_block_zn<false, true>(D, campos, ij, T, Tinv, dense_partials, tile, p, b, red, valid_lists, valid_counts, zn_lut);
        }
    }
#ifdef BTBA_WG_TRACE
    if (D.wg_trace && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        unsigned long long *q = D.wg_trace + 4 * (size_t)g;
        q[0] = wg_t0; q[1] = wall_clock64(); q[2] = (unsigned long long)(hw & 0xFFFFu) | ((unsigned long long)(xcc & 0xFu) << 16) | ((is_sparse ? 0ull : (wg_dbg[2] & 0xFFFFull)) << 32);
        q[3] = is_sparse ? 1ull : (((wg_dbg[0] - wg_t0) & 0xFFFFFFull) << 8) | (((wg_dbg[1] - wg_t0) & 0xFFFFFFull) << 32);      // kind | prologue end | loop end (ticks from start)
#ifdef BTBA_TRIP_TRACE
        if (!is_sparse) { q[0] = wg_dbg[4]; q[1] = wg_dbg[5]; q[3] = wg_dbg[6] << 8; q[2] = (q[2] & 0xFFFFFFFFull) | (wg_dbg[7] << 32); }       // wave 0's phase sums instead of the timeline (scripts/trip_trace.py)
#endif
    }
#endif
}

#define BTBA_FUSED_ITEM_ARGS D, n_d, n_s, g, campos, normals, dense_pairs, T, Tinv, dense_partials, corr, pair_offsets, sparse_partials, valid_lists, valid_counts, red, zn_lut
template <int LAYOUT>
__global__ void __launch_bounds__(kBlock, BTBA_FUSED_WAVES) k_fused_sweeps(SolveDims D, unsigned n_d, unsigned n_s,
                                                           const float4 *__restrict__ campos, const float4 *__restrict__ normals,
                                                           const int2 *__restrict__ dense_pairs, const float *__restrict__ T, const float *__restrict__ Tinv,
                                                           float *__restrict__ dense_partials,
                                                           const float4 *__restrict__ corr, const uint32_t *__restrict__ pair_offsets, float *__restrict__ sparse_partials,
                                                           const uint32_t *__restrict__ valid_lists, const int *__restrict__ valid_counts)
{
    __shared__ float red[kRedFloats];
    extern __shared__ __attribute__((aligned(16))) float zn_lut[];        // (Wd + Hd) floats (+ the block walk's list), compact layouts only
    const unsigned g = blockIdx.x;
    fused_item<LAYOUT>(BTBA_FUSED_ITEM_ARGS);
}

#undef BTBA_FUSED_ITEM_ARGS

// ---- system solve -------------------------------------------------------------------------------
// symmetric 3x3 packed xx xy xz yy yz zz
__device__ __forceinline__ float sym3(const float *m, int r, int c) { if (r > c) { const int t = r; r = c; c = t; } return m[r * 3 - r * (r - 1) / 2 + (c - r)]; }
// [v]x (r, c)
__device__ __forceinline__ float skew(const float *v, int r, int c)
{
    if (r == c) return 0.0f;
    const int k = 3 - r - c;                          // the remaining axis
    const float s = ((c - r + 3) % 3 == 1) ? -1.0f : 1.0f;   // (0,1)->-z (1,2)->-x (2,0)->-y
    return s * v[k];
}

// 6x6 sparse block entries in [trans, rot] order from the per-pair moment sums.
//   J_k = [ I | D(w_k) ],  D(w) = -[w]x.
// diag block of endpoint e (0 = i, 1 = j):  [[ n I, -[s]x ], [ [s]x, tr(M) I - M ]]
__device__ __forceinline__ float sparse_diag_entry(const float *rec, int e, int r, int c)
{
    const float *s = rec + 1 + 3 * e, *M = rec + 7 + 6 * e;
    const int br = r / 3, bc = c / 3, rr = r % 3, cc = c % 3;
    if (br == 0 && bc == 0) return rr == cc ? rec[0] : 0.0f;
    if (br == 0 && bc == 1) return -skew(s, rr, cc);
    if (br == 1 && bc == 0) return skew(s, rr, cc);
    const float tr = M[0] + M[3] + M[5];
    return (rr == cc ? tr : 0.0f) - sym3(M, rr, cc);
}
// cross block J_i^T J_j (row index on frame i, column on frame j):
//   [[ n I, -[s_j]x ], [ [s_i]x, tr(Mij) I - Mij^T ]]
__device__ __forceinline__ float sparse_cross_entry(const float *rec, int r, int c)
{
    const float *si = rec + 1, *sj = rec + 4, *Mij = rec + 19;
    const int br = r / 3, bc = c / 3, rr = r % 3, cc = c % 3;
    if (br == 0 && bc == 0) return rr == cc ? rec[0] : 0.0f;
    if (br == 0 && bc == 1) return -skew(sj, rr, cc);
    if (br == 1 && bc == 0) return skew(si, rr, cc);
    const float tr = Mij[0] + Mij[4] + Mij[8];
    return (rr == cc ? tr : 0.0f) - Mij[cc * 3 + rr];
}

// Linear form of sparse_diag_entry / sparse_cross_entry: value = c1 rec[i1 + e es] + c2 rec[i2 + e es].
// Returned as int4 (i1 | es << 8, i2, bits(c1), bits(c2)); es = record offset per endpoint (0 for n and cross blocks).
// (c1, c2 are 0 or +-1; the table is built once per window size on the host, btba_api.hip: solve_tables)
__host__ __device__ inline void sparse_entry_descriptor(bool cross, int r, int c, int out[4])
{
    const int br = r / 3, bc = c / 3, rr = r % 3, cc = c % 3;
    int i1 = 0, i2 = 0, es = 0;
    int c1 = 0, c2 = 0;                                                // signs
    const int k = 3 - rr - cc;
    const int sgn = ((cc - rr + 3) % 3 == 1) ? -1 : 1;                 // skew(v, rr, cc) = sgn v[k] for rr != cc
    if (br == 0 && bc == 0) { if (rr == cc) { i1 = 0; c1 = 1; } }
    else if (br == 0 && bc == 1) { if (rr != cc) { i1 = (cross ? 4 : 1) + k; c1 = -sgn; es = cross ? 0 : 3; } }
    else if (br == 1 && bc == 0) { if (rr != cc) { i1 = 1 + k; c1 = sgn; es = cross ? 0 : 3; } }
    else if (rr == cc) {                                               // tr(M) - M_rr = the other two diagonal entries
        const int a = (rr + 1) % 3, b = (rr + 2) % 3;
        const int dg[3] = { 0, 3, 5 };
        i1 = cross ? 19 + 4 * a : 7 + dg[a]; i2 = cross ? 19 + 4 * b : 7 + dg[b]; c1 = 1; c2 = 1; es = cross ? 0 : 6;
    } else {
        const int lo = rr < cc ? rr : cc, hi = rr < cc ? cc : rr;
        i1 = cross ? 19 + cc * 3 + rr : 7 + lo * 3 - lo * (lo - 1) / 2 + (hi - lo); c1 = -1; es = cross ? 0 : 6;
    }
    const int one = 0x3F800000, minus_one = (int)0xBF800000u;          // fp32 bit patterns of +-1
    out[0] = i1 | (es << 8); out[1] = i2;
    out[2] = c1 > 0 ? one : c1 < 0 ? minus_one : 0; out[3] = c2 > 0 ? one : c2 < 0 ? minus_one : 0;
}

__device__ __forceinline__ float block_sum(float v, float *scratch)
{
    // all threads must call; returns the workgroup total to every thread
    const float s = wave_sum_to_lane63(v);
    const int nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) scratch[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = scratch[0];
    for (int w = 1; w < nw; w++) t += scratch[w];
    return t;
}

// sum of q[0], q[stride], ... (count terms) in index order, four loads issued before the adds
__device__ __forceinline__ float strided_sum(const float *__restrict__ q, int count, int stride)
{
    float s = 0.0f;
    int c = 0;
    for (; c + 4 <= count; c += 4) {
        const float a0 = q[(size_t)c * stride], a1 = q[(size_t)(c + 1) * stride], a2 = q[(size_t)(c + 2) * stride], a3 = q[(size_t)(c + 3) * stride];
        s += a0; s += a1; s += a2; s += a3;
    }
    for (; c < count; c++) s += q[(size_t)c * stride];
    return s;
}

// ---- large windows (matrix in a global scratch): reduction and assembly on MANY workgroups -------------------------------
// Above BTBA_MAX_FRAMES_LDS frames k_system_solve used to do everything on one workgroup; at N = 85 (3 570 pairs, a 510 x 510
// system) 118 k of its 989 k cycles went into reducing the sweep partials and 562 k into the assembly -- chains of dependent
// global-memory reads walked by one CU (scripts/sys_clocks_big.py).  Both are embarrassingly parallel over pairs / matrix entries:
//   k_big_reduce    one thread per reduced value: pair sums = sum over chunks / tiles of the sweep partials, in partial order
//   k_big_assemble  one thread per off-diagonal entry, EIGHT lanes per diagonal entry / per unknown of the right-hand side (each sums an
//                   eighth of the frame's pairs in fixed order, fixed shuffle tree), plus the zero-fill of what nobody writes
// k_system_solve then starts at the PCG (D.pre_assembled).  Same formulas as phases A and B of k_system_solve; the sums over a
// frame's pairs are grouped in eighths instead of halves / quarters.  System layout per instance: A[n][ld], rhs[ld], prec[ld].
__global__ void __launch_bounds__(256) k_big_reduce(SolveDims D, float *sparse_partials, float *dense_partials,       // (not const: accumulators in BTBA_REDUCE_ATOMIC mode, cleared here)
                                                    float *__restrict__ pairsum_global)
{
    const int b = blockIdx.y;
    const size_t ns = (size_t)D.n_pairs * kSparseVals, nd = (size_t)D.n_dense_pairs * kDenseVals;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= ns + nd) return;
    float *out = pairsum_global + (size_t)b * (ns + nd);
    float s = 0.0f;
    if (e < ns) {
        if (D.use_sparse) {
            const size_t p = e / kSparseVals, v = e - p * kSparseVals;
            float *q = sparse_partials + (size_t)b * D.sp_stride + (p * D.sparse_chunks) * kSparseVals + v;
            for (int c = 0; c < D.sparse_chunks; c++) s += q[(size_t)c * kSparseVals];
            if (D.atomic_sums) q[0] = 0.0f;
        }
    } else if (D.use_dense) {
        const size_t ed = e - ns, p = ed / kDenseVals, v = ed - p * kDenseVals;
        float *q = dense_partials + (size_t)b * D.dp_stride + (p * D.dense_tiles) * kDenseVals + v;
        for (int c = 0; c < D.dense_tiles; c++) s += q[(size_t)c * kDenseVals];
        if (D.atomic_sums) q[0] = 0.0f;
    }
    out[e] = s;
}

// task segments of k_big_assemble, each padded to a multiple of 64 threads
struct BigTasks { unsigned off_diag, diag, rhs, zero, total; };
__host__ __device__ inline BigTasks big_tasks(int N, int P, int ld)
{
    const unsigned n = 6u * (unsigned)N;
    auto pad = [](unsigned x) { return (x + 63u) & ~63u; };
    BigTasks t;
    t.off_diag = pad((unsigned)P * 36u);
    t.diag = pad((unsigned)(N - 1) * 36u * 8u);
    t.rhs = pad(n * 8u);
    t.zero = pad(6u * (unsigned)ld + (n - 6u) * (6u + (unsigned)ld - n));
    t.total = t.off_diag + t.diag + t.rhs + t.zero;
    return t;
}

__global__ void __launch_bounds__(256) k_big_assemble(SolveDims D, const float *__restrict__ pairsum_global, const int2 *__restrict__ dense_pairs,
                                                      const int *__restrict__ adj_off, const int *__restrict__ adj, const int *__restrict__ solve_tab,
                                                      float *__restrict__ A_scratch)
{
    const int b = blockIdx.y;
    const int N = D.n_frames, n = 6 * N, ld = 4 * (((n + 3) / 4) | 1);
    const BigTasks K = big_tasks(N, D.n_pairs, ld);
    unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= K.total) return;
    float *A = A_scratch + (size_t)b * (size_t)(n + 2) * ld, *rhs_out = A + (size_t)n * ld, *prec_out = rhs_out + ld;
    const float *ps = pairsum_global + (size_t)b * ((size_t)D.n_pairs * kSparseVals + (size_t)D.n_dense_pairs * kDenseVals);
    const float *pd = ps + (size_t)D.n_pairs * kSparseVals;
    const int *pair_ij = solve_tab, *entry_lut = solve_tab + D.n_pairs;
    const int n_adj = D.use_dense ? 2 * D.n_dense_pairs : 0;
    const int *cross_tab = adj + n_adj;
    if (t < K.off_diag) {                                  // ---- off-diagonal entries, one canonical pair (i < j) each
        if (t >= (unsigned)D.n_pairs * 36u) return;
        const int p = (int)(t / 36u), rc = (int)(t - 36u * (unsigned)p), r = rc / 6, c = rc - 6 * r;
        const int pij = pair_ij[p], i = pij >> 8, j = pij & 255;
        if (i == 0) return;
        float v = 0.0f;
        if (D.use_sparse) {
            const int *dl = entry_lut + 4 * (36 + rc);
            const float *rec = ps + (size_t)p * kSparseVals;
            v = -D.w_sparse * (__int_as_float(dl[2]) * rec[dl[0] & 255] + __int_as_float(dl[3]) * rec[dl[1] & 255]);
        }
        const int dp = D.use_dense ? cross_tab[p] : -1;
        if (dp >= 0) v -= pd[(size_t)dp * kDenseVals + tri21(r, c)];
        A[(size_t)(6 * i + r) * ld + 6 * j + c] = v;
        A[(size_t)(6 * j + c) * ld + 6 * i + r] = v;
        return;
    }
    t -= K.off_diag;
    if (t < K.diag) {                                      // ---- diagonal blocks: 8 lanes per (frame k >= 1, entry)
        const unsigned e = t >> 3, sub = t & 7u;
        const bool live = e < (unsigned)(N - 1) * 36u;
        float v = 0.0f;
        int k = 1, rc = 0;
        if (live) {
            k = 1 + (int)(e / 36u); rc = (int)(e % 36u);
            const int r = rc / 6, c = rc - 6 * r;
            if (D.use_sparse) {
                const int *dl = entry_lut + 4 * rc;
                const int d0 = dl[0], d1 = dl[1], es = d0 >> 8;
                const float c1 = __int_as_float(dl[2]), c2 = __int_as_float(dl[3]);
                float acc = 0.0f;
                for (int m = (N * (int)sub) / 8; m < (N * ((int)sub + 1)) / 8; m++) {
                    if (m == k) continue;
                    const int i = m < k ? m : k, j = m < k ? k : m;
                    const float *rec = ps + (size_t)pair_index(i, j, N) * kSparseVals + (k == i ? 0 : es);
                    acc += c1 * rec[d0 & 255] + c2 * rec[d1 & 255];
                }
                v = D.w_sparse * acc;
            }
            if (D.use_dense) {
                const int t21 = tri21(r, c);
                const int q0 = adj_off[k], nq = adj_off[k + 1] - q0;
                float acc = 0.0f;
                for (int q = q0 + (nq * (int)sub) / 8; q < q0 + (nq * ((int)sub + 1)) / 8; q++) acc += pd[(size_t)(adj[q] >> 1) * kDenseVals + t21];
                v += acc;
            }
        }
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        if (live && sub == 0) A[(size_t)(6 * k + rc / 6) * ld + 6 * k + rc % 6] = v;
        return;
    }
    t -= K.diag;
    if (t < K.rhs) {                                       // ---- right-hand side and Jacobi diagonal: 8 lanes per unknown
        const unsigned e = t >> 3, sub = t & 7u;
        const bool live = e < (unsigned)n;
        const int k = live ? (int)e / 6 : 0, r = live ? (int)e % 6 : 0;
        float rhs = 0.0f, md = 0.0f;
        if (live && k > 0) {
            if (D.use_sparse) {
                const int o_i = (r < 3) ? 28 + r : 31 + r - 3, o_j = (r < 3) ? 28 + r : 34 + r - 3;
                const int p_i = (r < 3) ? 37 : 38 + r - 3, p_j = (r < 3) ? 37 : 41 + r - 3;
                for (int m = (N * (int)sub) / 8; m < (N * ((int)sub + 1)) / 8; m++) {
                    if (m == k) continue;
                    const int i = m < k ? m : k, j = m < k ? k : m;
                    const float *rec = ps + (size_t)pair_index(i, j, N) * kSparseVals;
                    const bool is_i = (k == i);
                    const float g = rec[is_i ? o_i : o_j];
                    rhs += is_i ? -g : g;
                    md += rec[is_i ? p_i : p_j];
                }
                rhs *= D.w_sparse;
            }
            if (D.use_dense) {
                const int q0 = adj_off[k], nq = adj_off[k + 1] - q0;
                float jtr = 0.0f;
                for (int q = q0 + (nq * (int)sub) / 8; q < q0 + (nq * ((int)sub + 1)) / 8; q++) {
                    const int a = adj[q];
                    const float g = pd[(size_t)(a >> 1) * kDenseVals + 21 + r];
                    jtr += (a & 1) ? g : -g;
                }
                rhs -= jtr;
            }
        }
        rhs += __shfl_xor(rhs, 1, 64); md += __shfl_xor(md, 1, 64);
        rhs += __shfl_xor(rhs, 2, 64); md += __shfl_xor(md, 2, 64);
        rhs += __shfl_xor(rhs, 4, 64); md += __shfl_xor(md, 4, 64);
        if (live && sub == 0) {
            rhs_out[e] = rhs;
            prec_out[e] = (k > 0) ? ((md > kEps) ? 1.0f / md : 1.0f) : 0.0f;
        }
        return;
    }
    t -= K.rhs;
    {                                                      // ---- what nobody writes: frame 0's rows and columns, the pad columns
        const unsigned z0 = 6u * (unsigned)ld, w = 6u + (unsigned)ld - (unsigned)n;
        if (t < z0) A[t] = 0.0f;
        else if (t - z0 < (unsigned)(n - 6) * w) {
            const unsigned q = t - z0, row = 6u + q / w, col = q % w;
            A[(size_t)row * ld + (col < 6u ? col : (unsigned)n + col - 6u)] = 0.0f;
        }
    }
}

// What one instance's system solve reads and writes, every pointer already at THIS instance.  The plain launch (k_system_solve) updates x, T
// and T^-1 in place; the chained launch (k_chain) reads iterate `it` and writes iterate `it + 1` into the next slot of a ring, because the
// sweep workgroups of other instances -- and of this instance's next iteration -- are running at the same time.
struct SolveIO {
    float *sparse_partials, *dense_partials;            // this iteration's sweep partials (not const: in BTBA_REDUCE_ATOMIC mode they are accumulators that the solve clears)
    const float *x_in, *T_in;                           // this iterate
    float *x_out, *T_out, *Tinv_out;                    // the next one
    float *poses_out;                                   // last iteration: the caller's pose buffer (or nullptr)
    float *pairsum_global;                              // reduced pair sums when they are not staged in LDS
    float *A_scratch;                                   // A_GLOBAL / pre_assembled / CHAIN: A[n][ld], rhs[ld], prec[ld]
    float *trace;                                       // this instance's trace records (or nullptr)
    unsigned long long *stamps;                         // developer (scripts/chain_trace.py): 8 wall-clock stamps of the phases of this solve (or nullptr)
};

// The system solve of ONE instance by the calling workgroup (all of its threads must call; dynamic LDS: A[n*ld] (unless A_GLOBAL / CHAIN)
// + 7 vectors[n] + scratch + tables + (optionally) reduced pair sums).
//   CHAIN = false: k_system_solve's body, a workgroup of 1 024 threads, one-wave PCG on the LDS-resident matrix (or the 16-wave PCG of A_GLOBAL).
//   CHAIN = true : the same sums in the same order run by a 256-thread workgroup of k_chain that shares its compute unit with five sweep
//                  workgroups and may only use a sweep workgroup's share of the LDS: the matrix lives in an L2-resident global scratch, the
//                  mat-vec of the PCG is spread over all four waves (two lanes per row, each the two partial sums of the one-wave version
//                  that belong to its half of the 16-byte chunks), everything else of a PCG step is the one-wave code, bit for bit.
template <bool LDS_PAIRS, bool A_GLOBAL, bool CHAIN, int NTHR>
__device__ __forceinline__ void system_solve_body(const SolveDims &D, int iter, const int tid, float *lds, const SolveIO &io,
                                                  const int2 *__restrict__ dense_pairs, const int *__restrict__ adj_off, const int *__restrict__ adj,
                                                  const int *__restrict__ solve_tab)
{
    constexpr int nthr = NTHR;                           // the workgroup size, a compile-time constant: loop strides and bounds against it fold
    float *sparse_partials = io.sparse_partials, *dense_partials = io.dense_partials;
    float *A_scratch = io.A_scratch;
    const int N = D.n_frames, n = 6 * N, ld = 4 * (((n + 3) / 4) | 1);   // odd multiple of 4: 16-B rows, conflict-free ds_read_b128
    // windows of more than BTBA_MAX_FRAMES_LDS frames: the matrix does not fit the CU's LDS and lives in an L2-resident
    // global scratch (A_GLOBAL); everything else keeps its place
    constexpr bool A_IN_GLOBAL = A_GLOBAL || CHAIN;
    float *A = A_IN_GLOBAL ? A_scratch : lds;      // global layout per instance: A[n][ld], rhs[ld], prec[ld]
    float *vb = A_IN_GLOBAL ? lds : A + (size_t)n * ld;       // rhs / residual r   (A's pad columns n..ld-1 stay 0)
    float *vM = vb + ld, *vz = vM + ld, *vp = vz + ld, *vAp = vp + ld, *vd = vAp + ld;     // vector stride ld (16-B aligned, zero padded)
    float *scratch = vd + ld;             // 16 floats
    float *vT = scratch + 16;             // this iterate's T[N][16]
    int *dense_pairs_lds = reinterpret_cast<int *>(vT + 16 * N);   // (target, source) of every dense pair: 2 Pd ints
    int *adj_off_l = dense_pairs_lds + 2 * D.n_dense_pairs, *adj_l = adj_off_l + (N + 1);      // adjacency staged in LDS
    int *pair_ij_l = adj_l + 2 * D.n_dense_pairs;                  // canonical pair p -> (i << 8 | j): P ints
    // every entry of a sparse 6x6 block is  c1 rec[i1 + e es] + c2 rec[i2 + e es]  of the pair's 44 moment sums (e = which
    // end of the pair): 36 descriptors for diagonal blocks + 36 for cross blocks, so the assembly loops are branch-free
    int *entry_lut = pair_ij_l + D.n_pairs;                        // 72 x 4 ints
    // reduced pair sums: in LDS when they fit (address space known at compile time -> ds_* instructions, not flat_*),
    // otherwise in an L2-resident global scratch (K = 30)
    float *ps;
    int *cross_l = entry_lut + 288;                                // canonical pair -> the dense pair whose cross block it carries (or -1): P ints, padded to 16 bytes
    float *x_l = reinterpret_cast<float *>(cross_l + ((D.n_pairs + 3) & ~3));       // this iterate's x (6 N floats, padded to 16 bytes): phase D would otherwise start with a fabric-latency load
    // CHAIN: a workgroup of k_chain owns a sweep workgroup's share of the LDS (a sixth of the compute unit's 160 KB).  What is left of it behind
    // the tables and vectors -- region R -- holds the reduced SPARSE pair sums while the system is assembled (two of their values go into every
    // matrix entry per pair; the dense sums, one value per entry and pair, stay in the global scratch and are fetched in batches) and afterwards
    // the assembled matrix, packed: the lower triangle of its (n - 6) x (n - 6) non-zero part.  The host checks that both fit (chain_lds_floats).
    float *region_R = lds + ((((x_l + ((6 * N + 3) & ~3)) - lds) + 3) & ~3);      // 16-byte aligned (the tables in front of it are not multiples of four words)
    if (CHAIN) ps = region_R;
    else if (LDS_PAIRS) ps = x_l + ((6 * N + 3) & ~3);
    else ps = io.pairsum_global;
    float *pd = CHAIN ? io.pairsum_global : ps + (size_t)D.n_pairs * kSparseVals;          // model-frame dense pair sums (S, g, count)
    float *tr = (D.trace_on && io.trace) ? io.trace + (size_t)iter * D.trace_record : nullptr;

    const long long clk0 = tr ? (long long)clock64() : 0;
#define BTBA_STAMP(slot) do { if (tr && tid == 0) tr[D.tr_clk + (slot)] = (float)((long long)clock64() - clk0); \
                              if (CHAIN && io.stamps && tid == 0) io.stamps[slot] = (unsigned long long)wall_clock64(); } while (0)
    if (D.pre_assembled) {
        // larger windows: k_big_reduce + k_big_assemble have built A, the right-hand side and the Jacobi diagonal in the global scratch;
        // a matrix that fits LDS is loaded back for the one-wave PCG (coalesced, once)
        const float *Ag = A_scratch;
        const float *rhs_g = Ag + (size_t)n * ld, *prec_g = rhs_g + ld;
        if (!A_IN_GLOBAL) {
            const float4 *src = reinterpret_cast<const float4 *>(Ag);
            float4 *dst = reinterpret_cast<float4 *>(A);
            for (int e = tid; e < (n * ld) / 4; e += nthr) dst[e] = src[e];
        }
        for (int e = tid; e < 16 * N; e += nthr) vT[e] = io.T_in[e];
        for (int e = tid; e < 6 * N; e += nthr) x_l[e] = io.x_in[e];
        for (int e = tid; e < n; e += nthr) { vb[e] = rhs_g[e]; vM[e] = prec_g[e]; vd[e] = 0.0f; }
        for (int e = n + tid; e < ld; e += nthr) vp[e] = 0.0f;
        __syncthreads();
    } else {
    // Staging of this iterate's T and of the pair tables into LDS.  Every one of these global loads is a fabric-latency
    // miss (~2 k cycles; written by the previous launch / the host), and a load -> LDS-store loop per array serialises
    // them behind s_waitcnt (measured: 7.7 k cycles for four tiny arrays).  So: the first trip of all four arrays is
    // loaded into registers here, the partial sums' loads are issued behind them, and the LDS stores happen after
    // the reduction -- the staging latency hides behind the first round of partial loads.
    const int n_dp = D.n_dense_pairs, n_adj = D.use_dense ? 2 * D.n_dense_pairs : 0, n_ao = D.use_dense ? N + 1 : 0;
    const float st_T = tid < 16 * N ? io.T_in[tid] : 0.0f;
    const float st_x = tid < 6 * N ? io.x_in[tid] : 0.0f;
    const int2 st_dp = tid < n_dp ? dense_pairs[tid] : make_int2(0, 0);
    const int st_ao = tid < n_ao ? adj_off[tid] : 0;
    const int st_adj = tid < n_adj ? adj[tid] : 0;
    // canonical pair -> (i << 8 | j) and the 72 entry descriptors: constant per window size, tabulated by the host
    const int st_pij = tid < D.n_pairs ? solve_tab[tid] : 0;
    const int st_lut = tid < 288 ? solve_tab[D.n_pairs + tid] : 0;
    const int *cross_tab = adj + n_adj;                            // behind the adjacency in the dense pair table (host: solve_enqueue)
    const int st_cross = (D.use_dense && tid < D.n_pairs) ? cross_tab[tid] : -1;
    // Phase A: fixed-order reduction of the sweep partials
    BTBA_STAMP(6);
    // several items per lane per trip, up to eight partials of each in flight (24-40 loads issued before the first add): the
    // partials were written by other XCDs' workgroups, every load comes from the fabric side of this XCD's L2 at ~1-2 k
    // cycles, so the phase costs (number of dependent load rounds) x (that latency) -- 6 rounds at B = 1 (10 tiles,
    // 5 chunks) instead of the 14 of a 2-item x 4-partial scheme.  Sums in partial order (fixed), whatever the grouping.
    auto reduce_partials = [&](auto vals_c, auto items_c, float *src, float *dst, int n_items_pairs, int parts) {
        constexpr int vals = decltype(vals_c)::value;
        constexpr int kItems = decltype(items_c)::value;
        const int total = n_items_pairs * vals;
        for (int e0 = tid; e0 < total; e0 += kItems * nthr) {
            float *q[kItems];
            float sum[kItems];
            bool own[kItems];                                              // dead slots re-read the lane's first item (and must not clear it twice)
#pragma unroll
            for (int i = 0; i < kItems; i++) {
                const int e = e0 + i * nthr;
                const int ec = e < total ? e : e0;
                own[i] = e < total;
                q[i] = src + (size_t)(ec / vals) * parts * vals + (ec % vals);
                sum[i] = 0.0f;
            }
            int c = 0;
            auto round = [&](auto width_c) {
                constexpr int width = decltype(width_c)::value;
                float v[kItems][width];
#pragma unroll
                for (int i = 0; i < kItems; i++)
#pragma unroll
                    for (int u = 0; u < width; u++) v[i][u] = q[i][(size_t)(c + u) * vals];
#pragma unroll
                for (int i = 0; i < kItems; i++)
#pragma unroll
                    for (int u = 0; u < width; u++) sum[i] += v[i][u];
                if (D.atomic_sums) {                    // the records are accumulators: leave them empty for the next iteration's sweeps
#pragma unroll
                    for (int i = 0; i < kItems; i++)
#pragma unroll
                        for (int u = 0; u < width; u++) if (own[i]) q[i][(size_t)(c + u) * vals] = 0.0f;
                }
                c += width;
            };
            if (!CHAIN) {                               // (the chained launch runs with <= 2 chunks / tiles: kChainMaxParts, checked by the host)
                while (c + 8 <= parts) round(std::integral_constant<int, 8>{});
                if (c + 4 <= parts) round(std::integral_constant<int, 4>{});
            }
            if (c + 2 <= parts) round(std::integral_constant<int, 2>{});
            if (c < parts) round(std::integral_constant<int, 1>{});
#pragma unroll
            for (int i = 0; i < kItems; i++) { const int e = e0 + i * nthr; if (e < total) dst[e] = sum[i]; }
        }
    };
    // items per lane per trip: one trip covers the c3 window (105 pairs: 4 620 sparse / 2 940 dense sums over 1 024 lanes).  (Reducing both kinds at
    // the same time on disjoint groups of waves -- one round of fabric-latency loads instead of two -- measured 0.5 us SLOWER per launch, r03 call 39.)
    // (k_chain: 256 lanes under the sweep's 80-register budget and at most two partials per sum -- more items per lane, narrower rounds)
    // Batches big enough to fill the chip run with one or two partials per sum (pick_chunks, pick_tiles): 16-byte loads then, and ALL of a lane's loads --
    // sparse and dense records -- issued before the first add: ONE round of fabric latency instead of the two (plain kernel: sparse, then dense) or four
    // (256-lane chained solve) of the scalar scheme below.  The same sums: 0 + first (+ second).  Chained solve 6.4 -> 4.5 us; plain kernel: see DESIGN.md 4.3.
    if (CHAIN || (D.sparse_chunks <= 2 && D.dense_tiles <= 2 && !D.atomic_sums)) {
        constexpr int kS4 = kSparseVals / 4, kD4 = kDenseVals / 4, kRounds = CHAIN ? 6 : 2;          // sums of four a lane holds per pass: 15 frames = 1 155 + 735 of them, 7.4 per lane of 256, 1.85 per lane of 1 024
        const int ns4 = D.use_sparse ? D.n_pairs * kS4 : 0, nd4 = D.use_dense ? D.n_dense_pairs * kD4 : 0;
        if (!D.use_sparse) for (int e = tid; e < D.n_pairs * kSparseVals; e += nthr) ps[e] = 0.0f;
        for (int e0 = tid; e0 < ns4 + nd4; e0 += kRounds * nthr) {
            float4 v0[kRounds], v1[kRounds];
            bool two[kRounds];
#pragma unroll
            for (int u = 0; u < kRounds; u++) {
                const int e = min(e0 + u * nthr, ns4 + nd4 - 1);
                const bool sp = e < ns4;
                const int q = sp ? e : e - ns4, per = sp ? kS4 : kD4, parts = sp ? D.sparse_chunks : D.dense_tiles;
                const int rec = q / per, k = q - rec * per;
                const float4 *src = reinterpret_cast<const float4 *>(sp ? sparse_partials : dense_partials) + ((size_t)rec * parts) * per + k;
                two[u] = parts > 1;
                v0[u] = src[0];
                v1[u] = src[two[u] ? per : 0];
            }
#pragma unroll
            for (int u = 0; u < kRounds; u++) {
                const int e = e0 + u * nthr;
                if (e >= ns4 + nd4) continue;
                float4 sum = make_float4(0.0f + v0[u].x, 0.0f + v0[u].y, 0.0f + v0[u].z, 0.0f + v0[u].w);
                if (two[u]) { sum.x += v1[u].x; sum.y += v1[u].y; sum.z += v1[u].z; sum.w += v1[u].w; }
                // (four scalar stores: the plain kernel's LDS copy of the pair sums is not 16-byte aligned; two branches, not a selected pointer: ps and pd
                // may live in different address spaces, and a select between them sends this compiler's backend into an illegal instruction)
                if (e < ns4) { float *dst = ps + 4 * (size_t)e; dst[0] = sum.x; dst[1] = sum.y; dst[2] = sum.z; dst[3] = sum.w; }
                else { float *dst = pd + 4 * (size_t)(e - ns4); dst[0] = sum.x; dst[1] = sum.y; dst[2] = sum.z; dst[3] = sum.w; }
            }
        }
    } else {
    if (D.use_sparse) reduce_partials(std::integral_constant<int, kSparseVals>{}, std::integral_constant<int, 5>{}, sparse_partials, ps, D.n_pairs, D.sparse_chunks);
    else for (int e = tid; e < D.n_pairs * kSparseVals; e += nthr) ps[e] = 0.0f;
    if (D.use_dense) reduce_partials(std::integral_constant<int, kDenseVals>{}, std::integral_constant<int, 3>{}, dense_partials, pd, D.n_dense_pairs, D.dense_tiles);
    }
    BTBA_STAMP(7);
    // the staged values: first trip from the registers loaded above, the rest (windows beyond 64 frames / 1 024 pairs) by loops
    if (tid < 16 * N) vT[tid] = st_T;
    if (tid < 6 * N) x_l[tid] = st_x;
    for (int e = tid + nthr; e < 6 * N; e += nthr) x_l[e] = io.x_in[e];
    if (tid < D.n_pairs) pair_ij_l[tid] = st_pij;
    if (tid < 288) entry_lut[tid] = st_lut;
    for (int e = tid + nthr; e < 288; e += nthr) entry_lut[e] = solve_tab[D.n_pairs + e];
    if (tid < D.n_pairs) cross_l[tid] = st_cross;
    for (int e = tid + nthr; e < D.n_pairs; e += nthr) { pair_ij_l[e] = solve_tab[e]; cross_l[e] = D.use_dense ? cross_tab[e] : -1; }
    if (tid < n_dp) { dense_pairs_lds[2 * tid] = st_dp.x; dense_pairs_lds[2 * tid + 1] = st_dp.y; }
    if (tid < n_ao) adj_off_l[tid] = st_ao;
    for (int e = tid + nthr; e < n_ao; e += nthr) adj_off_l[e] = adj_off[e];
    if (tid < n_adj) adj_l[tid] = st_adj;
    for (int e = tid + nthr; e < 16 * N; e += nthr) vT[e] = io.T_in[e];
    for (int e = tid + nthr; e < n_dp; e += nthr) { const int2 ij = dense_pairs[e]; dense_pairs_lds[2 * e] = ij.x; dense_pairs_lds[2 * e + 1] = ij.y; }
    for (int e = tid + nthr; e < n_adj; e += nthr) adj_l[e] = adj[e];
    // zero what the assembly below does not write: frame 0's rows and columns (it stays fixed) and the pad columns n .. ld-1 that the
    // 16-byte mat-vec chunks read; every other entry is assigned by phase B (all canonical pairs with i >= 1, all diagonal blocks k >= 1)
    if (!CHAIN) {                                       // (the chained solve packs the non-zero part of the matrix and never reads the rest)
        for (int e = tid; e < 6 * ld; e += nthr) A[e] = 0.0f;
        for (int e = tid; e < (n - 6) * (6 + ld - n); e += nthr) {
            const int row = 6 + e / (6 + ld - n), q = e % (6 + ld - n);
            A[row * ld + (q < 6 ? q : n + q - 6)] = 0.0f;
        }
    }
    __syncthreads();
    BTBA_STAMP(0);
    // (the camera-frame -> model-frame congruence of the dense pair sums is done by the sweep workgroups: dense_epilogue)
    if (tr && D.use_dense) for (int e = tid; e < D.n_dense_pairs * kDenseVals; e += nthr) tr[D.tr_dpair + e] = pd[e];

    BTBA_STAMP(1);
    // Phase B1: off-diagonal 6x6 blocks, one canonical pair (i<j) each: A_ij = -(ws Ji^T Jj + S_dense).  The dense part comes from the
    // dense pair listed as (target i, source j) for this canonical pair, if any (cross_l; a pair listed the other way round is erased
    // by the reference's FlipJtJ, duplicates of a pair in an explicit list are not supported -- documented): ONE pass, no
    // read-modify-write, no barrier before the diagonal blocks.
    // CHAIN: the dense pair sums come from the global scratch -- an L2 round trip each, and a store to A between two of them keeps the compiler
    // from issuing the next before the previous has returned (61 us for this phase, profiles/r04/chain_experiments.json).  So the loads of a
    // batch of entries are issued first and consumed afterwards: the same values enter the same sums in the same order.
    // sum of the dense pair sums `off` of adjacency entries [qa, qb) of a frame, signed by the frame's role in the pair when `by_role`
    auto dense_adj_sum = [&](int qa, int qb, int off, bool by_role) {
        float acc = 0.0f;
        for (int q = qa; q < qb; q += 8) {
            float v[8];
            int a[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { a[u] = adj_l[min(q + u, qb - 1)]; v[u] = pd[(size_t)(a[u] >> 1) * kDenseVals + off]; }
#pragma unroll
            for (int u = 0; u < 8; u++) if (q + u < qb) acc += by_role ? ((a[u] & 1) ? v[u] : -v[u]) : v[u];
        }
        return acc;
    };
    if (CHAIN) {
        constexpr int kB = 4;                             // entries per lane and batch
        const int total = D.n_pairs * 36;
        for (int e0 = tid; e0 < total; e0 += kB * nthr) {
            int pp[kB], rcv[kB], ij[kB];
            float dv[kB];
            bool has[kB];
#pragma unroll
            for (int u = 0; u < kB; u++) {
                const int e = e0 + u * nthr, ec = min(e, total - 1);
                pp[u] = ec / 36; rcv[u] = ec - 36 * pp[u];
                const int pij = pair_ij_l[pp[u]];
                ij[u] = (e < total && (pij >> 8) != 0) ? pij : 0;             // 0: nothing to do (beyond the end, or frame 0's rows / columns)
                const int dp = ij[u] ? cross_l[pp[u]] : -1;
                has[u] = dp >= 0;
                dv[u] = has[u] ? pd[(size_t)dp * kDenseVals + tri21(rcv[u] / 6, rcv[u] % 6)] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kB; u++) {
                if (!ij[u]) continue;
                const int p = pp[u], rc = rcv[u], r = rc / 6, c = rc - 6 * r, i = ij[u] >> 8, j = ij[u] & 255;
                float v = 0.0f;
                if (D.use_sparse) {
                    const int *dl = entry_lut + 4 * (36 + rc);
                    const int4 d = make_int4(dl[0], dl[1], dl[2], dl[3]);
                    const float *rec = ps + (size_t)p * kSparseVals;
                    v = -D.w_sparse * (__int_as_float(d.z) * rec[d.x & 255] + __int_as_float(d.w) * rec[d.y & 255]);
                }
                if (has[u]) v -= dv[u];
                A[(6 * j + c) * ld + 6 * i + r] = v;       // (the lower-triangle copy is the one the packing reads)
            }
        }
    } else
    for (int e = tid; e < D.n_pairs * 36; e += nthr) {
        const int p = e / 36, rc = e - 36 * p, r = rc / 6, c = rc - 6 * r;
        const int pij = pair_ij_l[p];
        const int i = pij >> 8, j = pij & 255;
        if (i == 0) continue;
        float v = 0.0f;
        if (D.use_sparse) {
            const int *dl = entry_lut + 4 * (36 + rc);
            const int4 d = make_int4(dl[0], dl[1], dl[2], dl[3]);
            const float *rec = ps + (size_t)p * kSparseVals;
            v = -D.w_sparse * (__int_as_float(d.z) * rec[d.x & 255] + __int_as_float(d.w) * rec[d.y & 255]);
        }
        const int dp = cross_l[p];
        if (dp >= 0) v -= pd[(size_t)dp * kDenseVals + tri21(r, c)];
        A[(6 * i + r) * ld + 6 * j + c] = v;
        A[(6 * j + c) * ld + 6 * i + r] = v;
    }
    // Phase B2: diagonal blocks: TWO lanes per (frame k >= 1, entry), each sums half of the frame's pairs in fixed order
    for (int t = tid; t < 2 * (N - 1) * 36; t += nthr) {
        const int e = t >> 1, half = t & 1;
        const int k = 1 + e / 36, r = (e % 36) / 6, c = e % 6;
        float v = 0.0f;
        if (D.use_sparse) {
            const int *dl = entry_lut + 4 * (e % 36);
            const int4 d = make_int4(dl[0], dl[1], dl[2], dl[3]);
            const int es = d.x >> 8;
            const float c1 = __int_as_float(d.z), c2 = __int_as_float(d.w);
            const int m0 = half ? N / 2 : 0, m1 = half ? N : N / 2;
            float acc = 0.0f;
            for (int m = m0; m < m1; m++) {
                if (m == k) continue;
                const int i = m < k ? m : k, j = m < k ? k : m;
                const float *rec = ps + (size_t)pair_index(i, j, N) * kSparseVals + (k == i ? 0 : es);
                acc += c1 * rec[d.x & 255] + c2 * rec[d.y & 255];
            }
            v = D.w_sparse * acc;
        }
        if (D.use_dense) {
            const int t21 = tri21(r, c);
            const int q0 = adj_off_l[k], q1 = adj_off_l[k + 1], qm = q0 + (q1 - q0) / 2;
            float acc = 0.0f;
            if (CHAIN) acc = dense_adj_sum(half ? qm : q0, half ? q1 : qm, t21, false);
            else for (int q = half ? qm : q0; q < (half ? q1 : qm); q++) acc += pd[(size_t)(adj_l[q] >> 1) * kDenseVals + t21];
            v += acc;
        }
        v += __shfl_xor(v, 1, 64);                       // partner lane = same entry, other half (t and t^1 share a wave)
        if (!half) A[(6 * k + r) * ld + 6 * k + c] = v;
    }
    // right-hand side and Jacobi diagonal: FOUR lanes per unknown, each sums a quarter of the frame's pairs
    for (int t = tid; t < 4 * n; t += nthr) {
        const int e = t >> 2, part = t & 3;
        const int k = e / 6, r = e % 6;
        float rhs = 0.0f, md = 0.0f;
        if (k > 0) {
            if (D.use_sparse) {
                const int m0 = (N * part) / 4, m1 = (N * (part + 1)) / 4;
                const int o_i = (r < 3) ? 28 + r : 31 + r - 3, o_j = (r < 3) ? 28 + r : 34 + r - 3;       // rhs slots of the i / j end
                const int p_i = (r < 3) ? 37 : 38 + r - 3, p_j = (r < 3) ? 37 : 41 + r - 3;              // preconditioner slots
                for (int m = m0; m < m1; m++) {
                    if (m == k) continue;
                    const int i = m < k ? m : k, j = m < k ? k : m;
                    const float *rec = ps + (size_t)pair_index(i, j, N) * kSparseVals;
                    const bool is_i = (k == i);
                    const float g = rec[is_i ? o_i : o_j];
                    rhs += is_i ? -g : g;
                    md += rec[is_i ? p_i : p_j];
                }
                rhs *= D.w_sparse;
            }
            if (D.use_dense) {
                const int q0 = adj_off_l[k], nq = adj_off_l[k + 1] - q0;
                float jtr = 0.0f;
                if (CHAIN) jtr = dense_adj_sum(q0 + (nq * part) / 4, q0 + (nq * (part + 1)) / 4, 21 + r, true);
                else for (int q = q0 + (nq * part) / 4; q < q0 + (nq * (part + 1)) / 4; q++) {
                    const int a = adj_l[q];
                    const float g = pd[(size_t)(a >> 1) * kDenseVals + 21 + r];
                    jtr += (a & 1) ? g : -g;            // source frame: row_j = a;  target frame: row_i = -a
                }
                rhs -= jtr;
            }
        }
        rhs += __shfl_xor(rhs, 1, 64); md += __shfl_xor(md, 1, 64);        // the four lanes of an unknown sit in one wave
        rhs += __shfl_xor(rhs, 2, 64); md += __shfl_xor(md, 2, 64);
        if (part == 0) {
            vb[e] = rhs;
            vM[e] = (k > 0) ? ((md > kEps) ? 1.0f / md : 1.0f) : 0.0f;
            vd[e] = 0.0f;
        }
    }
    for (int e = n + tid; e < ld; e += nthr) vp[e] = 0.0f;      // pad of p: read by the 16-byte mat-vec chunks
    __syncthreads();
    }
    BTBA_STAMP(2);
    if (CHAIN) {
        // the matrix into region R (the sparse pair sums are spent): entry (a, c), c <= a, of its non-zero (n - 6) x (n - 6) part at a (a + 1) / 2 + c.
        // A is symmetric bit for bit (off-diagonal blocks are one value for both positions; a diagonal block's entries (r, c) and (c, r) are the
        // same sums of the same values), so the packed lower triangle holds every value a row of the mat-vec reads.
        const int na = n - 6, n_tri = na * (na + 1) / 2;
        for (int e0 = tid; e0 < n_tri; e0 += 8 * nthr) {       // eight loads per lane in flight (the copy used to run one L2 round trip per element: 17 us)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = min(e0 + u * nthr, n_tri - 1);
                int a = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);       // row of packed index e, fixed up for the rounding of the root
                a -= (a * (a + 1) / 2 > e) ? 1 : 0;
                a += ((a + 1) * (a + 2) / 2 <= e) ? 1 : 0;
                const int c = e - a * (a + 1) / 2;
                v[u] = A[(size_t)(6 + a) * ld + 6 + c];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) if (e0 + u * nthr < n_tri) region_R[e0 + u * nthr] = v[u];
        }
        __syncthreads();
    }
    if (tr) {
        for (int e = tid; e < n; e += nthr) {
            const int k = e / 6, r = e % 6;           // trace order (rot, trans); internal [trans, rot]
            const int o = k * 6 + (r < 3 ? r + 3 : r - 3);
            tr[D.tr_rhs + o] = vb[e];
            tr[D.tr_prec + o] = vM[e];
        }
        for (int e = tid; e < n * n; e += nthr) tr[D.tr_A + e] = A[(e / n) * ld + (e % n)];
    }

    BTBA_STAMP(5);
    // Phase C: Jacobi-preconditioned CG, SolverBundling.cu:575-818 (frame 0 entries stay 0).
    // ONE wave runs it: <= 186 unknowns = <= 3 rows per lane (4 provisioned), vectors live in registers, the two dot products per
    // step are DPP wave sums, only p travels through LDS.  No workgroup barrier inside the iteration (the
    // 16-wave version spent ~6 k cycles per step in barriers; this one ~1 k).
    if (A_GLOBAL) {
        // large windows (n up to 510): all 16 waves, one row per thread.  A is symmetric, so thread `row` walks COLUMN `row`
        // (A[k ld + row]: neighbouring threads read neighbouring addresses), p and the other vectors stay in LDS, the two dot
        // products per step are workgroup sums.  Same recurrences and epsilon guards as the single-wave version below.
        const int row = tid;
        const bool live = row < n;
        float r_ = live ? vb[row] : 0.0f, m_ = live ? vM[row] : 0.0f, d_ = 0.0f;
        float p_ = m_ * r_;
        if (live) vp[row] = p_;
        float rz = block_sum(r_ * p_, scratch);          // (contains the barrier that publishes vp)
        for (int li = 0; li < D.n_pcg; li++) {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            if (live) {
                int k = 0;
                for (; k + 4 <= n; k += 4) {
                    a0 += A[(size_t)k * ld + row] * vp[k]; a1 += A[(size_t)(k + 1) * ld + row] * vp[k + 1];
                    a2 += A[(size_t)(k + 2) * ld + row] * vp[k + 2]; a3 += A[(size_t)(k + 3) * ld + row] * vp[k + 3];
                }
                for (; k < n; k++) a0 += A[(size_t)k * ld + row] * vp[k];
            }
            const float ap = (a0 + a1) + (a2 + a3);
            const float pAp = block_sum(live ? p_ * ap : 0.0f, scratch);
            const float alpha = (pAp > kEps) ? rz / pAp : 0.0f;
            d_ = d_ + alpha * p_;
            r_ = r_ - alpha * (live ? ap : 0.0f);
            const float z_ = m_ * r_;
            const float rz_new = block_sum(z_ * r_, scratch);
            const float beta = (rz > kEps) ? rz_new / rz : 0.0f;
            if (tr && tid == 0) { float *sc = tr + D.tr_pcg + 4 * li; sc[0] = pAp; sc[1] = alpha; sc[2] = rz_new; sc[3] = beta; }
            rz = rz_new;
            p_ = z_ + beta * p_;
            __syncthreads();                             // every thread has finished reading the old p
            if (live) vp[row] = p_;
            __syncthreads();
        }
        if (live) vd[row] = d_;
    } else if (CHAIN || tid < 64) {
        // rows per lane: 2 up to 128 unknowns (windows of <= 21 frames: every tracker-size window), 4 up to 186 (N <= BTBA_MAX_FRAMES_LDS = 31).
        // One wave alone on its SIMD pays ~8 cycles per instruction, so the two dead rows of a 4-row loop cost real time (the sums only
        // ever add their exact zeros: same bits either way).
        auto pcg = [&](auto rows_c) {
        constexpr int kMaxRows = decltype(rows_c)::value;
        // CHAIN: wave 0 runs this code as it stands, except that the matrix-vector product of a step is computed by all four waves (the
        // matrix is in the global scratch: two rounds of L2-latency loads spread over 256 lanes instead of 23 chunks x 2 rows on one wave)
        const int lane = tid & 63;
        const bool w0 = tid < 64;
        float r_[kMaxRows], m_[kMaxRows], p_[kMaxRows], d_[kMaxRows];
        float part = 0.0f, rz = 0.0f;
        if (w0) {
#pragma unroll
            for (int j = 0; j < kMaxRows; j++) {
                const int row = lane + 64 * j;
                const bool live = row < n;
                r_[j] = live ? vb[row] : 0.0f; m_[j] = live ? vM[row] : 0.0f; d_[j] = 0.0f;
                p_[j] = m_[j] * r_[j];
                part += r_[j] * p_[j];
                if (live) vp[row] = p_[j];
            }
            rz = wave_sum_all(part);
        }
        for (int li = 0; li < D.n_pcg; li++) {
            float ap_[kMaxRows];
#pragma unroll
            for (int j = 0; j < kMaxRows; j++) ap_[j] = 0.0f;
            if (CHAIN) {
                // Row r of A p in the one-wave code below: four partial sums over the x, y, z, w lanes of its 16-byte chunks, fused multiply-adds
                // in chunk order, then (x + y) + (z + w).  Here lane pair (2 q, 2 q + 1) owns row 6 + q (frame 0's rows are zero: their product is
                // an exact +0), lane h of the pair runs the two sums of its half of every chunk, and one exchange adds the halves: the same
                // operations on the same operands in the same order.
                // The matrix is the packed triangle in region R: A[row][k] = L[a (a + 1) / 2 + c] for c <= a, L[c (c + 1) / 2 + a] above the diagonal
                // (a = row - 6, c = k - 6).  Columns 0 .. 5 (frame 0) and the pad columns hold exact zeros and zeros of p: skipped, their
                // fused multiply-adds leave the sums as they are.
                __syncthreads();                              // this step's p is in LDS
                const int h = tid & 1, na = n - 6;
                const float *L = region_R;
                for (int pr = tid >> 1; pr < na; pr += nthr >> 1) {
                    const int row = 6 + pr, a = pr, row_base = a * (a + 1) / 2;
                    float sa = 0.0f, sb = 0.0f;
                    // chunk cc covers columns 4 cc .. 4 cc + 3; this lane its columns k0 = 4 cc + 2 h and k0 + 1
                    int c0 = 4 + 2 * h - 6;                                 // active column of k0 in chunk 1 (chunk 0 lies inside frame 0's zero columns)
                    int tri0 = c0 * (c0 + 1) / 2, tri1 = (c0 + 1) * (c0 + 2) / 2;      // c (c + 1) / 2 of the two columns, advanced by 4 c + 10 per chunk
                    for (int k0 = 4 + 2 * h; k0 < n; k0 += 4) {
                        const float2 q = *reinterpret_cast<const float2 *>(vp + k0);
                        if (c0 >= 0) sa = __builtin_fmaf(L[c0 <= a ? row_base + c0 : tri0 + a], q.x, sa);
                        if (c0 + 1 >= 0 && c0 + 1 < na) sb = __builtin_fmaf(L[c0 + 1 <= a ? row_base + c0 + 1 : tri1 + a], q.y, sb);
                        tri0 += 4 * c0 + 10; tri1 += 4 * c0 + 14;
                        c0 += 4;
                    }
                    const float th = sa + sb;
                    const float ap = th + __shfl_xor(th, 1, 64);
                    if (h == 0) vAp[row] = ap;
                }
                if (tid < 6) vAp[tid] = 0.0f;
                __syncthreads();
                if (w0) {
#pragma unroll
                    for (int j = 0; j < kMaxRows; j++) ap_[j] = vAp[min(lane + 64 * j, n - 1)];
                }
            } else {
            const float4 *a0 = reinterpret_cast<const float4 *>(A + (size_t)min(lane, n - 1) * ld), *a1 = reinterpret_cast<const float4 *>(A + (size_t)min(lane + 64, n - 1) * ld);
            const float4 *a2 = reinterpret_cast<const float4 *>(A + (size_t)min(lane + 128, n - 1) * ld), *a3 = reinterpret_cast<const float4 *>(A + (size_t)min(lane + 192, n - 1) * ld);
            const float4 *p4 = reinterpret_cast<const float4 *>(vp);
            const int nq = ld >> 2;                       // 16-byte chunks per row (pad columns hold zeros)
            // four independent partial sums per row (x, y, z, w lanes of the 16-byte chunks): the single wave has no other
            // work to hide FMA latency behind, so a 92-long dependent chain per row was the critical path of this phase
            // ... as PACKED fp32 FMAs (v_pk_fma_f32, two lanes of a chunk per instruction): this wave is alone on its SIMD, so it is
            // bound by the one-instruction-per-four-cycles issue rate, where a packed FMA costs what a plain one does
            // (profiles/r02/valu_calibration.md) -- half the instructions for the same fused multiply-adds, bit for bit
            typedef float f2 __attribute__((ext_vector_type(2)));
            auto lo = [](const float4 &v) { return (f2){ v.x, v.y }; };
            auto hi = [](const float4 &v) { return (f2){ v.z, v.w }; };
            f2 q0a = (f2){ 0.f, 0.f }, q0b = q0a, q1a = q0a, q1b = q0a, q2a = q0a, q2b = q0a, q3a = q0a, q3b = q0a;
            if (kMaxRows == 2) {
_Pragma("unroll 8")
                for (int c = 0; c < nq; c++) {
                    const float4 pc = p4[c], r0 = a0[c], r1 = a1[c];
                    q0a = __builtin_elementwise_fma(lo(r0), lo(pc), q0a); q0b = __builtin_elementwise_fma(hi(r0), hi(pc), q0b);
                    q1a = __builtin_elementwise_fma(lo(r1), lo(pc), q1a); q1b = __builtin_elementwise_fma(hi(r1), hi(pc), q1b);
                }
            } else {
_Pragma("unroll 4")
                for (int c = 0; c < nq; c++) {
                    const float4 pc = p4[c], r0 = a0[c], r1 = a1[c], r2 = a2[c], r3 = a3[c];
                    q0a = __builtin_elementwise_fma(lo(r0), lo(pc), q0a); q0b = __builtin_elementwise_fma(hi(r0), hi(pc), q0b);
                    q1a = __builtin_elementwise_fma(lo(r1), lo(pc), q1a); q1b = __builtin_elementwise_fma(hi(r1), hi(pc), q1b);
                    q2a = __builtin_elementwise_fma(lo(r2), lo(pc), q2a); q2b = __builtin_elementwise_fma(hi(r2), hi(pc), q2b);
                    q3a = __builtin_elementwise_fma(lo(r3), lo(pc), q3a); q3b = __builtin_elementwise_fma(hi(r3), hi(pc), q3b);
                }
            }
            ap_[0] = (q0a.x + q0a.y) + (q0b.x + q0b.y); ap_[1] = (q1a.x + q1a.y) + (q1b.x + q1b.y);
            if (kMaxRows == 4) { ap_[kMaxRows - 2] = (q2a.x + q2a.y) + (q2b.x + q2b.y); ap_[kMaxRows - 1] = (q3a.x + q3a.y) + (q3b.x + q3b.y); }
            }
            if (!w0) continue;
            part = 0.0f;
#pragma unroll
            for (int j = 0; j < kMaxRows; j++) part += (lane + 64 * j < n) ? p_[j] * ap_[j] : 0.0f;
            const float pAp = wave_sum_all(part);
            const float alpha = (pAp > kEps) ? rz / pAp : 0.0f;
            float z_[kMaxRows];
            part = 0.0f;
#pragma unroll
            for (int j = 0; j < kMaxRows; j++) {
                const bool live = lane + 64 * j < n;
                d_[j] = d_[j] + alpha * p_[j];
                r_[j] = r_[j] - alpha * (live ? ap_[j] : 0.0f);
                z_[j] = m_[j] * r_[j];
                part += z_[j] * r_[j];
            }
            const float rz_new = wave_sum_all(part);
            const float beta = (rz > kEps) ? rz_new / rz : 0.0f;
            if (tr && lane == 0) { float *sc = tr + D.tr_pcg + 4 * li; sc[0] = pAp; sc[1] = alpha; sc[2] = rz_new; sc[3] = beta; }
            rz = rz_new;
#pragma unroll
            for (int j = 0; j < kMaxRows; j++) {
                p_[j] = z_[j] + beta * p_[j];
                if (lane + 64 * j < n) vp[lane + 64 * j] = p_[j];
            }
        }
        if (w0) {
#pragma unroll
            for (int j = 0; j < kMaxRows; j++) if (lane + 64 * j < n) vd[lane + 64 * j] = d_[j];
        }
        };
        if (n <= 128) pcg(std::integral_constant<int, 2>{}); else pcg(std::integral_constant<int, 4>{});
    }
    __syncthreads();

    BTBA_STAMP(3);
    // Phase D: x_k <- Log(Exp(delta_k) Exp(x_k)); next iterate's T, T^-1  (SolverBundling.cu:805-815, 890-897)
    for (int k = tid; k < N; k += nthr) {
        float *xk = (CHAIN ? io.x_out : const_cast<float *>(io.x_in)) + 6 * k;       // (in place unless chained: one base pointer less to keep)
        const float *xl = x_l + 6 * k;
        float rot[3] = { xl[0], xl[1], xl[2] }, trans[3] = { xl[3], xl[4], xl[5] };
        if (k > 0) {
            const float dW[3] = { vd[6 * k + 3], vd[6 * k + 4], vd[6 * k + 5] }, dT[3] = { vd[6 * k], vd[6 * k + 1], vd[6 * k + 2] };
            const Mat4 U = pose_to_matrix(dW, dT);
            const Mat4 C = load_mat4(vT + 16 * k);      // = Exp(x_k): this iterate's T, computed from the same x_k by the previous launch
            matrix_to_pose(mat_mul(U, C), rot, trans);
        }
        if (k > 0 || CHAIN) { xk[0] = rot[0]; xk[1] = rot[1]; xk[2] = rot[2]; xk[3] = trans[0]; xk[4] = trans[1]; xk[5] = trans[2]; }      // (in place: frame 0 keeps its x)
        const Mat4 E = pose_to_matrix(rot, trans);
        store_mat4((CHAIN ? io.T_out : const_cast<float *>(io.T_in)) + 16 * k, E);
        if (io.poses_out) store_mat4(io.poses_out + 16 * k, E);      // last iterate: convertPosesToMatricesCU (SBA.cpp:115), no separate copy
        store_mat4(io.Tinv_out + 16 * k, mat_inverse(E));
        if (tr) {
            for (int q = 0; q < 3; q++) { tr[D.tr_x + 6 * k + q] = rot[q]; tr[D.tr_x + 6 * k + 3 + q] = trans[q]; }
            for (int q = 0; q < 16; q++) tr[D.tr_T + 16 * k + q] = E.m[q];
            for (int q = 0; q < 3; q++) { tr[D.tr_delta + 6 * k + q] = vd[6 * k + 3 + q]; tr[D.tr_delta + 6 * k + 3 + q] = vd[6 * k + q]; }
        }
    }
    BTBA_STAMP(4);
#undef BTBA_STAMP
}

// grid (B) x 1 024: one workgroup per instance, x / T / T^-1 updated in place.
template <bool LDS_PAIRS, bool A_GLOBAL = false>
__global__ void __launch_bounds__(kSolveBlock) k_system_solve(SolveDims D, int iter,
                                                        float *sparse_partials, float *dense_partials,
                                                        const int2 *__restrict__ dense_pairs, const int *__restrict__ adj_off, const int *__restrict__ adj,
                                                        float *__restrict__ x, float *__restrict__ T, float *__restrict__ Tinv,
                                                        float *__restrict__ pairsum_global, float *__restrict__ trace, float *__restrict__ A_scratch = nullptr,
                                                        float *__restrict__ poses_out = nullptr, const int *__restrict__ solve_tab = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const size_t b = blockIdx.x;
    const int N = D.n_frames, n = 6 * N, ld = 4 * (((n + 3) / 4) | 1);
    SolveIO io;
    io.sparse_partials = sparse_partials + b * D.sp_stride; io.dense_partials = dense_partials + b * D.dp_stride;
    io.x_in = io.x_out = x + b * D.x_stride;
    io.T_in = io.T_out = T + b * D.pose_stride; io.Tinv_out = Tinv + b * D.pose_stride;
    io.poses_out = poses_out ? poses_out + 16 * b * N : nullptr;
    io.pairsum_global = pairsum_global ? pairsum_global + b * ((size_t)D.n_pairs * kSparseVals + (size_t)D.n_dense_pairs * kDenseVals) : nullptr;
    io.A_scratch = A_scratch ? A_scratch + b * (size_t)(n + 2) * ld : nullptr;
    io.trace = trace ? trace + b * (size_t)D.n_gn * D.trace_record : nullptr;
    io.stamps = nullptr;
    system_solve_body<LDS_PAIRS, A_GLOBAL, false, kSolveBlock>(D, iter, (int)threadIdx.x, lds, io, dense_pairs, adj_off, adj, solve_tab);
}


// ---- the chained launch: ALL Gauss-Newton iterations of a batch in ONE launch --------------------------------------------------------------
// The plain schedule is 2 launches per iteration: the fused sweeps (every compute unit busy, then 12-15 % of the launch draining at falling
// occupancy), then k_system_solve (one workgroup per instance: 32 of 256 compute units for ~18 us, VALU busy 0.04) -- seven times per solve, the
// reference's serial tail (PCGInit / PCGStep*, SolverBundling.cu:575-818, 931-1003) in miniature.  Instances are independent, so nothing
// forces instance b's iteration i + 1 to wait for instance c's iteration i.  k_chain puts the work items of all iterations into one grid:
//
//   per XCD x (blocks g = 8 s + x, s = position in the XCD's sequence; the dispatcher places block g on XCD g % 8 and starts an XCD's blocks
//   in order -- observed, used for speed and for forward progress, never for the values computed) and per iteration, for each instance the XCD
//   owns:  [its dense items, heaviest pairs first] [its sparse items] [ONE solve item]
//
//   sweep item (iteration i, instance b): waits until flags[b] >= i (iterate i of b is published: one relaxed agent-scope load per wave, long
//       true in the steady state), does exactly what its k_fused_sweeps workgroup does -- same partial record, bit for bit -- stores the record
//       write-through, drains its stores and adds 1 to arrivals[i][b].
//   solve item (i, b): waits until arrivals[i][b] counts all of b's sweep items, takes an agent-scope acquire, runs system_solve_body<CHAIN>
//       (k_system_solve's sums in k_system_solve's order on 256 threads) from ring slot i into ring slot i + 1, releases at agent scope and
//       sets flags[b] = i + 1.
//
// While instance b is being solved, the compute unit's other five workgroup slots -- and the rest of the chip -- run the sweep items of the
// XCD's other instances; b's next items come up in the sequence (inst_per_xcd - 1) instances later.  No launch boundary, no drain, no idle
// 224 compute units, except once at the very end.
// Forward progress: an item only ever waits for items EARLIER in its own XCD's sequence (all items of an instance sit on one XCD), which are
// resident or finished when it starts; every wait is bounded by a watchdog that raises *error and lets the launch run out (the host then
// reports the solve as failed) -- a misbehaving dispatcher costs a wrong answer that is flagged, never a hung GPU.
// Visibility (MI355X_MICROARCH.md, "inter-workgroup visibility"): every hand-off goes to a region of its own -- ring slot / iteration /
// instance, padded to whole 128-byte lines -- that no workgroup touches before its producer has published it, so no cache can hold an older
// copy of it; producers write through (sweep records: sc1 stores + vmcnt drain) or release at agent scope (solve items) before the relaxed
// agent-scope counter / flag update; the solve item acquires at agent scope; sweep items read the poses through the scalar cache, which they
// invalidate after the wait, and through addresses the compiler cannot move above the wait.
constexpr int kChainMaxParts = 2;     // chunks / tiles per pair the chained solve item reduces (system_solve_body<CHAIN>)
constexpr size_t kChainLdsBytes = 26560;   // dynamic LDS of a k_chain workgroup: with the 736 static bytes a sixth of the compute unit's 160 KB -- six workgroups per CU, as k_fused_sweeps
// floats of region R a chained solve needs behind its tables and vectors (system_solve_body<CHAIN>): the reduced sparse pair sums, then the packed matrix
constexpr int kChainMaxFrames = 21;       // (128 lane pairs own the rows of the matrix: 6 N - 6 <= 128; the LDS budget stops at 15 frames before that)
__host__ __device__ inline size_t chain_region_floats(int n_frames) { const size_t P = (size_t)n_frames * (n_frames - 1) / 2, na = 6 * (size_t)n_frames - 6; return P * kSparseVals > na * (na + 1) / 2 ? P * kSparseVals : na * (na + 1) / 2; }
struct ChainDims {
    int n_iter;                 // Gauss-Newton iterations in this launch
    int n_inst;                 // B
    unsigned inst_per_xcd;      // ceil(B / 8): instance b lives on XCD b / inst_per_xcd
    unsigned items_d, items_s;  // sweep items per (instance, iteration): dense_tiles * Pd, sparse_chunks * P
    unsigned sparse_period;     // 0: an instance's sparse items follow its dense items; R >= 2: every R-th slot of the instance is a sparse item until they are used up
    unsigned group;             // instances per GROUP of an XCD's sequence (sparse_period = 0): [dense items of the group's instances][their sparse items][their solve
                                // items] -- the short sparse items then come in fewer, longer runs (1: after every instance; inst_per_xcd: once per iteration)
    int *flags;                 // [B] published iterate of every instance (0 = the k_prepare output): what a waiting wave polls (L2, never a cache above it)
    const int *published;       // [n_iter + 1][B][16]: word 0 of line (e, b) becomes 1 when iterate e of instance b is published -- one 64-byte line per word,
                                // read through the scalar cache: a line that says 1 was fetched after the publication, a line that says 0 may be stale (-> poll flags[b])
    int *arrivals;              // [n_iter][B]
    int *error;                 // != 0: a wait ran into the watchdog (host-visible memory)
    long long timeout_ticks;    // 100 MHz ticks (wall_clock64)
    size_t pose_ring, x_ring;   // floats between two ring slots of T / T^-1, of x
    size_t sp_ring, dp_ring;    // floats between two iterations' sweep partials
    size_t ps_size, A_size;     // floats per (iteration, instance) of the reduced-pair-sum scratch and of the matrix scratch
    unsigned long long *trace;  // developer: per block (start, end of wait, end, kind | it << 8 | b << 16 | hardware id << 32) in 100 MHz ticks, or nullptr
    unsigned long long *stamps; // developer: [n_iter][B][8] phase stamps of the solve items (behind the block records), or nullptr
    int solve_prio;             // s_setprio of a solve item's waves (0 = the default priority of every wave)
    int last_solve_external;    // 1: the solve items of the LAST iteration do nothing -- the host launches k_system_solve for it (16 waves per instance on an
                                // idle chip: ~18 us, against the ~90 us a 4-wave solve item on a busy compute unit would leave exposed at the end of the launch)
    int debug_skip;             // developer TIMING experiments (wrong results by construction): 1 sweep items do not wait, 2 sweep items do not arrive, 4 solve items do not wait, 8 solve items do nothing, 64 solve items do not publish (tests/test_gpu_chain.py: the watchdog)
};

// one wave waits until *word >= want: lane 0 polls with relaxed agent-scope loads (served by L2, never by this CU's L1), sleeping in between
__device__ __forceinline__ void chain_wait_ge(const int *word, int want, const ChainDims &Cn)
{
    const bool lane0 = (item_tid() & 63u) == 0u;
    auto poll = [&]() {
        int v = 0;
        if (lane0) v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __builtin_amdgcn_readfirstlane(v);
    };
    if (poll() >= want) return;
    const long long t0 = wall_clock64();
    for (;;) {
        __builtin_amdgcn_s_sleep(16);
        if (poll() >= want) return;
        if (wall_clock64() - t0 > Cn.timeout_ticks) {
            if (lane0) __hip_atomic_store(Cn.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
}

template <int LAYOUT>   // 1: pinhole compact cache, 8 x 8 block walk or strips; 3: the same walking the frames' valid-pixel lists
__global__ void __launch_bounds__(kBlock, BTBA_FUSED_WAVES) k_chain(SolveDims D, ChainDims Cn, const float4 *__restrict__ zn,
                                                    float *T, float *Tinv, float *x, float *dense_partials,
                                                    const float4 *__restrict__ corr, const uint32_t *__restrict__ pair_offsets, float *sparse_partials,
                                                    const uint32_t *__restrict__ valid_lists, const int *__restrict__ valid_counts,
                                                    const int2 *__restrict__ dense_pairs, const int *__restrict__ adj_off, const int *__restrict__ adj,
                                                    const int *__restrict__ solve_tab, float *pairsum_global, float *A_scratch, float *poses_out)
{
    __shared__ float red[kRedFloats];
    extern __shared__ __attribute__((aligned(16))) float dyn_lds[];       // a sweep item's look-up tables, or a solve item's vectors and tables
    const unsigned g = blockIdx.x, xcd = g & 7u, sg = g >> 3;
    const unsigned n_sweep = Cn.items_d + Cn.items_s, slots_per_inst = n_sweep + 1u, per_iter = Cn.inst_per_xcd * slots_per_inst;
    const unsigned it = sg / per_iter, s = sg - it * per_iter;
    unsigned k = s / slots_per_inst, r = s - k * slots_per_inst;
    if (Cn.group > 1u) {
        // groups of `group` instances (the host makes inst_per_xcd a multiple of it): position inside the group -> (instance, kind, index)
        const unsigned per_group = Cn.group * slots_per_inst, gi = s / per_group, rg = s - gi * per_group;
        unsigned kg, rr;
        if (rg < Cn.group * Cn.items_d) { kg = rg / Cn.items_d; rr = rg - kg * Cn.items_d; }
        else if (rg < Cn.group * n_sweep) { const unsigned q = rg - Cn.group * Cn.items_d; kg = q / Cn.items_s; rr = Cn.items_d + (q - kg * Cn.items_s); }
        else { kg = rg - Cn.group * n_sweep; rr = n_sweep; }
        k = gi * Cn.group + kg; r = rr;
    }
    const int b = (int)(xcd * Cn.inst_per_xcd + k);
    if (b >= Cn.n_inst) return;
    const unsigned tid = item_tid();
    const unsigned long long t_start = Cn.trace ? (unsigned long long)wall_clock64() : 0ull;
    unsigned long long t_wait = t_start;
    int kind;
    if (r < n_sweep) {
        // ---- sweep item
        bool is_sparse;
        unsigned idx;                                           // index among the instance's sparse / dense items
        if (Cn.sparse_period >= 2u) {
            const unsigned q = r / Cn.sparse_period, m = r - q * Cn.sparse_period;
            is_sparse = (m == Cn.sparse_period - 1u) && (q < Cn.items_s);
            idx = is_sparse ? q : r - min(q, Cn.items_s);
        } else {
            is_sparse = r >= Cn.items_d;
            idx = is_sparse ? r - Cn.items_d : r;
        }
        const float *T_it = T + (size_t)it * Cn.pose_ring, *Tinv_it = Tinv + (size_t)it * Cn.pose_ring;
        if (it > 0u) {
            // Is iterate `it` of instance b published?  Fast path: its `published` line through the scalar cache (a hit costs ~100 cycles; in the
            // steady state the solve item finished tens of microseconds ago).  The line is this (iterate, instance)'s alone and is only ever
            // fetched by this instance's items of this iteration, so a cached 1 cannot be older than the publication; a 0 may be, and is not
            // trusted: the wave then polls flags[b] in L2.  The poses are read through the scalar cache as well, each 64-byte matrix a line
            // that nobody fetches before this point -- nothing to invalidate -- and through addresses the compiler sees only AFTER the wait, so
            // that no load of a pose can be issued before the publication has been seen (constant-address-space loads may otherwise be hoisted).
            const int seen = (Cn.debug_skip & 1) ? 1 : *as_const(Cn.published + ((size_t)it * Cn.n_inst + b) * 16);
            if (seen == 0) chain_wait_ge(Cn.flags + b, (int)it, Cn);
        }
        asm volatile("" : "+s"(T_it), "+s"(Tinv_it) :: "memory");
        if (Cn.trace) t_wait = (unsigned long long)wall_clock64();
        kind = is_sparse ? 1 : 0;
        if (is_sparse) {
            const int chunk = (int)(idx % (unsigned)D.sparse_chunks), p = (int)(idx / (unsigned)D.sparse_chunks);
            sparse_block(D, corr, pair_offsets, T_it, sparse_partials + (size_t)it * Cn.sp_ring, chunk, p, b, red);
        } else {
            const int tile = D.tile_major ? (int)(idx / (unsigned)D.n_dense_pairs) : (int)(idx % (unsigned)D.dense_tiles);
            const int p = D.tile_major ? (int)(idx % (unsigned)D.n_dense_pairs) : (int)(idx / (unsigned)D.dense_tiles);      // a WORK POSITION (dense_work_item)
            float *dp_it = dense_partials + (size_t)it * Cn.dp_ring;
            if (LAYOUT == 1 && D.walk_blocks) dense_block_pinhole<2>(D, zn, p, T_it, Tinv_it, dp_it, tile, b, red, valid_lists, valid_counts, dyn_lds);
            else if (LAYOUT == 1) dense_block_pinhole<0>(D, zn, p, T_it, Tinv_it, dp_it, tile, b, red, valid_lists, valid_counts, dyn_lds);
            else dense_block_pinhole<1>(D, zn, p, T_it, Tinv_it, dp_it, tile, b, red, valid_lists, valid_counts, dyn_lds);
        }
        // the record was stored write-through by lanes of wave 0: once those stores have been acknowledged, count this item in
        if (tid < 64u && !(Cn.debug_skip & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0u) __hip_atomic_fetch_add(Cn.arrivals + (size_t)it * Cn.n_inst + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        // ---- solve item: iterate `it` -> iterate `it + 1` of instance b
        kind = 2;
        if ((Cn.debug_skip & 8) || (Cn.last_solve_external && (int)it == Cn.n_iter - 1)) return;
        if (tid < 64u && !(Cn.debug_skip & 4)) {
            chain_wait_ge(Cn.arrivals + (size_t)it * Cn.n_inst + b, (int)n_sweep, Cn);
            if (tid == 0u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (Cn.trace) t_wait = (unsigned long long)wall_clock64();
        const int N = D.n_frames;
        const size_t ib = (size_t)it * Cn.n_inst + b;
        SolveIO io;
        io.sparse_partials = sparse_partials + (size_t)it * Cn.sp_ring + (size_t)b * D.sp_stride;
        io.dense_partials = dense_partials + (size_t)it * Cn.dp_ring + (size_t)b * D.dp_stride;
        io.x_in = x + (size_t)it * Cn.x_ring + (size_t)b * D.x_stride; io.x_out = x + (size_t)(it + 1u) * Cn.x_ring + (size_t)b * D.x_stride;
        io.T_in = T + (size_t)it * Cn.pose_ring + (size_t)b * D.pose_stride;
        io.T_out = T + (size_t)(it + 1u) * Cn.pose_ring + (size_t)b * D.pose_stride;
        io.Tinv_out = Tinv + (size_t)(it + 1u) * Cn.pose_ring + (size_t)b * D.pose_stride;
        io.poses_out = ((int)it == Cn.n_iter - 1) ? poses_out + 16 * (size_t)b * N : nullptr;
        io.pairsum_global = pairsum_global + ib * Cn.ps_size;
        io.A_scratch = A_scratch + ib * Cn.A_size;
        io.trace = nullptr;
        io.stamps = Cn.stamps ? Cn.stamps + 8 * ib : nullptr;
        // a solve item is a chain of short dependent phases on ONE workgroup that shares its compute unit with five sweep workgroups: every
        // instruction of it otherwise queues behind theirs; its instances' next items wait for it, theirs for nothing
        if (Cn.solve_prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (Cn.solve_prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (Cn.solve_prio == 3) __builtin_amdgcn_s_setprio(3);
        system_solve_body<false, false, true, kBlock>(D, (int)it, (int)tid, dyn_lds, io, dense_pairs, adj_off, adj, solve_tab);
        __syncthreads();
        if (tid == 0u && !(Cn.debug_skip & 64)) {           // (64: the watchdog's test -- an iterate that is never published)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the compiler may drop the wait behind the write-back: MI355X_MICROARCH.md, compiler hazard)
            __hip_atomic_store(Cn.flags + b, (int)it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(const_cast<int *>(Cn.published) + ((size_t)(it + 1u) * Cn.n_inst + b) * 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (Cn.trace && tid == 0u) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        unsigned long long *q = Cn.trace + 4 * (size_t)g;
        q[0] = t_start; q[1] = t_wait; q[2] = (unsigned long long)wall_clock64();
        q[3] = (unsigned long long)kind | ((unsigned long long)it << 8) | ((unsigned long long)b << 16) | ((unsigned long long)(hw & 0xFFFFu) << 32) | ((unsigned long long)(xcc & 0xFu) << 48);
    }
}

// after a chained solve: a raised watchdog word (host-visible memory) turns the solve's output poses into NaN
__global__ void __launch_bounds__(256) k_chain_poison(const int *__restrict__ error, float *__restrict__ poses, int n)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) poses[e] = __int_as_float(0x7FC00000);
}

// Log / Exp / inverse of the incoming matrices into ring slot 0 of the chained launch (k_prepare with padded instance strides)
__global__ void __launch_bounds__(64) k_prepare_strided(int total, int n_frames, int pose_stride, int x_stride, const float *__restrict__ poses,
                                                        float *__restrict__ x, float *__restrict__ T, float *__restrict__ Tinv)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = idx / n_frames, f = idx - b * n_frames;
    const Mat4 M = load_mat4(poses + 16 * (size_t)idx);
    float rot[3], trans[3];
    matrix_to_pose(M, rot, trans);
    float *o = x + (size_t)b * x_stride + 6 * f;
    o[0] = rot[0]; o[1] = rot[1]; o[2] = rot[2]; o[3] = trans[0]; o[4] = trans[1]; o[5] = trans[2];
    const Mat4 E = pose_to_matrix(rot, trans);
    store_mat4(T + (size_t)b * pose_stride + 16 * f, E);
    store_mat4(Tinv + (size_t)b * pose_stride + 16 * f, mat_inverse(E));
}

}  // namespace btba
