// btba_ransac.hpp -- correspondence RANSAC, the step before correspondences enter bundle adjustment
// (SURVEY.md 8(f) rank 4): 3-point rigid hypotheses, inlier vote, best trial, inlier list.
//   reference: ransacMultiPairGPU            src/cuda/cuda_ransac.cu:1228-1323
//              ransacEstimateModelKernel     :1145-1181   (sample + procrustesKernel :999-1102)
//              ransacEvalModelKernel         :1183-1200   (n_trials x n_pts threads, one atomicAdd per inlier,
//                                                          an n_trials x n_pts int flag matrix per pair)
//              findBestTrial                 :1202-1219   (atomicMax + "last writer wins")
//              host side                     SiftManager::runRansacMultiPairGPU, FeatureManager.cpp:659-741
// Design here: ONE vote launch for all pairs -- a lane owns a trial, builds its hypothesis in registers and walks the
// pair's points.  Two hypothesis builders: BTBA_RANSAC_REFERENCE_SVD (default) is the reference's procrustesKernel with its
// approximate 3x3 SVD restated operation for operation (btba_svd3.hpp) -- per-trial poses, inlier counts and the winner
// equal the reference's on identical sample triples; BTBA_RANSAC_HORN is the exact Kabsch optimum by Horn's closed form
// (the optimal rotation is the dominant eigenvector of a symmetric 4x4, found by cyclic Jacobi; no 3x3 SVD, no reflection
// case, no "R is not valid" failure; near-collinear samples rejected by the eigenvalue gap).  Points are staged through LDS in tiles
// that every lane reads at the same index (broadcast reads; no flag matrix, no float atomics); the best trial is an
// integer atomicMax on (count << 32 | ~trial): deterministic, lowest trial id among equals.  A second small
// launch re-evaluates the winning pose and writes the ordered inlier list (ballot compaction).  Sample triples: by default the
// reference's own -- its per-trial cuRAND XORWOW streams are one constant table of n_trials x 3 uniforms (btba_xorwow.hpp), which
// a lane turns into its pair's indices with a multiply and a round; or explicit; or a counter hash (distinct per pair).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "btba_svd3.hpp"

namespace btba {

__host__ __device__ __forceinline__ uint32_t mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
// index draw: round(u (n-1)), u in (0,1] -- the shape of round(curand_uniform() * (n_pts-1)), cuda_ransac.cu:1161-1163,
// from a counter hash instead of a cuRAND XORWOW stream (integer-exact, identical in oracle/btba_oracle_ransac.c)
__host__ __device__ __forceinline__ int ransac_draw(uint64_t seed, int pair, int trial, int draw, int n_pts)
{
#pragma clang fp contract(off)
    uint32_t h = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9E3779B9u * (uint32_t)(pair + 1)));
    h = mix32(h ^ (0x85EBCA6Bu * (uint32_t)(trial + 1)));
    h = mix32(h + 0xC2B2AE35u * (uint32_t)(draw + 1));
    const float u = (float)((h >> 8) + 1u) * (1.0f / 16777216.0f);
    return (int)roundf(u * (float)(n_pts - 1));
}

// round(curand_uniform(&state) * (n_pts - 1)), cuda_ransac.cu:1159-1161: int -> float, one fp32 multiply, round half away from zero
__host__ __device__ __forceinline__ int ransac_index(float u, int n_pts)
{
#pragma clang fp contract(off)
    return (int)roundf(u * (float)(n_pts - 1));
}

// ransacEvalModelKernel's test (:1190-1195): reject when |ptB - pose ptA| > thres.  Contraction off: the vote and
// the extraction kernel must round identically, or a point on the threshold would be counted but not listed.
__device__ __forceinline__ bool ransac_is_inlier(const float (&P)[12], const float4 a, const float4 b, float thres)
{
#pragma clang fp contract(off)
    const float dx = b.x - (P[0] * a.x + P[1] * a.y + P[2] * a.z + P[3] * a.w);
    const float dy = b.y - (P[4] * a.x + P[5] * a.y + P[6] * a.z + P[7] * a.w);
    const float dz = b.z - (P[8] * a.x + P[9] * a.y + P[10] * a.z + P[11] * a.w);
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    return !(dist > thres);
}

// Rigid motion src -> dst of n = 3 points (procrustesKernel, :999-1102): means and the 3x3 correlation S in the
// reference's order; rotation by Horn's method.  N(S) is the symmetric 4x4 whose dominant eigenvector is the unit
// quaternion of the optimal PROPER rotation (eigenvalues  s1+s2+d s3 > s1-s2-d s3 > ...), so the SVD's reflection
// branch does not exist; gap = (l1 - l2) / (l1 + l2) = (s2 + d s3) / s1 says how unique the optimum is.
__device__ __forceinline__ bool ransac_procrustes3(const float4 (&src)[3], const float4 (&dst)[3], float (&P)[12], float &gap)
{
    const float smx = (src[0].x + src[1].x + src[2].x) / 3, smy = (src[0].y + src[1].y + src[2].y) / 3, smz = (src[0].z + src[1].z + src[2].z) / 3;
    const float dmx = (dst[0].x + dst[1].x + dst[2].x) / 3, dmy = (dst[0].y + dst[1].y + dst[2].y) / 3, dmz = (dst[0].z + dst[1].z + dst[2].z) / 3;
    float S[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float s[3] = { src[i].x - smx, src[i].y - smy, src[i].z - smz }, d[3] = { dst[i].x - dmx, dst[i].y - dmy, dst[i].z - dmz };
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) S[r][c] += s[r] * d[c];
    }
    float A[4][4], V[4][4];
    A[0][0] = S[0][0] + S[1][1] + S[2][2];
    A[1][1] = S[0][0] - S[1][1] - S[2][2];
    A[2][2] = -S[0][0] + S[1][1] - S[2][2];
    A[3][3] = -S[0][0] - S[1][1] + S[2][2];
    A[0][1] = A[1][0] = S[1][2] - S[2][1];
    A[0][2] = A[2][0] = S[2][0] - S[0][2];
    A[0][3] = A[3][0] = S[0][1] - S[1][0];
    A[1][2] = A[2][1] = S[0][1] + S[1][0];
    A[1][3] = A[3][1] = S[2][0] + S[0][2];
    A[2][3] = A[3][2] = S[1][2] + S[2][1];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) V[r][c] = (r == c) ? 1.0f : 0.0f;
    // cyclic Jacobi, fixed 6 sweeps x 6 rotations (converges quadratically; 4 sweeps reach fp32 round-off)
    for (int sweep = 0; sweep < 6; sweep++) {
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                const float apq = A[p][q];
                const float theta = (A[q][q] - A[p][p]) / (2.0f * apq);
                float t = copysignf(1.0f, theta) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
                t = (fabsf(apq) > 1e-30f) ? t : 0.0f;                       // also swallows the 0/0 of an already-diagonal pair
                const float c = 1.0f / sqrtf(t * t + 1.0f), s = t * c;
                A[p][p] -= t * apq; A[q][q] += t * apq; A[p][q] = A[q][p] = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (r != p && r != q) {
                        const float arp = A[r][p], arq = A[r][q];
                        A[r][p] = A[p][r] = c * arp - s * arq;
                        A[r][q] = A[q][r] = s * arp + c * arq;
                    }
                    const float vrp = V[r][p], vrq = V[r][q];
                    V[r][p] = c * vrp - s * vrq;
                    V[r][q] = s * vrp + c * vrq;
                }
            }
    }
    // dominant eigenpair and the runner-up eigenvalue
    float l1 = A[0][0], l2 = -INFINITY;
    float qw = V[0][0], qx = V[1][0], qy = V[2][0], qz = V[3][0];
#pragma unroll
    for (int k = 1; k < 4; k++) {
        const float l = A[k][k];
        const bool better = l > l1;
        l2 = better ? l1 : fmaxf(l2, l);
        qw = better ? V[0][k] : qw; qx = better ? V[1][k] : qx; qy = better ? V[2][k] : qy; qz = better ? V[3][k] : qz;
        l1 = better ? l : l1;
    }
    gap = (l1 - l2) / (l1 + l2);
    const float inv = 1.0f / sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
    qw *= inv; qx *= inv; qy *= inv; qz *= inv;
    const float R[9] = { 1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                         2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                         2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy) };
#pragma unroll
    for (int r = 0; r < 3; r++) { P[4 * r] = R[3 * r]; P[4 * r + 1] = R[3 * r + 1]; P[4 * r + 2] = R[3 * r + 2]; }
    P[3] = dmx - (R[0] * smx + R[1] * smy + R[2] * smz);
    P[7] = dmy - (R[3] * smx + R[4] * smy + R[5] * smz);
    P[11] = dmz - (R[6] * smx + R[7] * smy + R[8] * smz);
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 12; k++) finite = finite && (fabsf(P[k]) < 1e30f);
    return finite && (l1 + l2 > 0.0f);
}

struct RansacDims {
    int n_pairs, n_trials;
    float dist_thres;
    uint64_t seed;
    int draw;            // where a trial's three indices come from: 0 counter hash of (seed, pair, trial, k); 1 explicit `samples`;
                         // 2 the restated cuRAND stream (unverified against a CUDA run, btba_xorwow.hpp): round(u_table[trial][k] * (n - 1)) (btba_xorwow.hpp)
    int hypothesis;      // BTBA_RANSAC_REFERENCE_SVD (0): procrustesKernel with the reference's approximate 3x3 SVD, operation for operation; BTBA_RANSAC_HORN (1)
};

// grid (ceil(n_trials / 256), n_pairs) x 256.  offsets[pair] .. offsets[pair+1] delimit the pair's points.
__global__ void __launch_bounds__(256) k_ransac_vote(RansacDims D, const float4 *__restrict__ ptsA, const float4 *__restrict__ ptsB, const int *__restrict__ offsets,
                                                    const int *__restrict__ samples, const float *__restrict__ u_table, float *__restrict__ poses,
                                                    int *__restrict__ counts, unsigned long long *__restrict__ best)
{
    const int pair = blockIdx.y, trial = blockIdx.x * 256 + (int)threadIdx.x;
    const int o = offsets[pair], n = offsets[pair + 1] - o;
    const float4 *A = ptsA + o, *B = ptsB + o;
    int cnt = 0;
    bool good = false;
    float P[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    if (trial < D.n_trials && n >= 3) {
        int idx[3];
#pragma unroll
        for (int k = 0; k < 3; k++)
            idx[k] = D.draw == 1 ? samples[3 * ((size_t)pair * D.n_trials + trial) + k]
                   : D.draw == 2 ? ransac_index(u_table[3 * trial + k], n) : ransac_draw(D.seed, pair, trial, k, n);
        const bool distinct = !(idx[0] == idx[1] || idx[1] == idx[2] || idx[0] == idx[2]);
        const bool in_range = idx[0] >= 0 && idx[1] >= 0 && idx[2] >= 0 && idx[0] < n && idx[1] < n && idx[2] < n;
        if (distinct && in_range) {
            const float4 s[3] = { A[idx[0]], A[idx[1]], A[idx[2]] }, d[3] = { B[idx[0]], B[idx[1]], B[idx[2]] };
            if (D.hypothesis == 0) {
                good = svd3::procrustes_reference(s, d, 3, P);              // false <=> the reference's "R is not valid" (P = identity)
            } else {
                float gap;
                good = ransac_procrustes3(s, d, P, gap) && gap >= 1e-4f;    // (near-)collinear samples do not define a motion
            }
        }
    }
    // the pair's points go through LDS in tiles of 256 (coalesced 16-byte loads by the whole workgroup); every lane then
    // reads the SAME point -- an LDS broadcast, conflict-free -- and tests it against its own hypothesis.  (First version:
    // uniform-address scalar loads straight from global memory; the dependent s_load latency made it 3-4x slower.)
    __shared__ float4 tileA[256], tileB[256];
    const bool any_good = __syncthreads_or(good ? 1 : 0) != 0;
    if (any_good)
        for (int base = 0; base < n; base += 256) {
            const int i_ld = min(base + (int)threadIdx.x, n - 1);
            tileA[threadIdx.x] = A[i_ld];
            tileB[threadIdx.x] = B[i_ld];
            __syncthreads();
            const int m = min(256, n - base);
#pragma unroll 4
            for (int i = 0; i < m; i++) cnt += ransac_is_inlier(P, tileA[i], tileB[i], D.dist_thres) ? 1 : 0;
            __syncthreads();
        }
    cnt = good ? cnt : 0;
    if (trial < D.n_trials) {
        counts[(size_t)pair * D.n_trials + trial] = cnt;
        float4 *out = reinterpret_cast<float4 *>(poses + 12 * ((size_t)pair * D.n_trials + trial));
        out[0] = make_float4(P[0], P[1], P[2], P[3]); out[1] = make_float4(P[4], P[5], P[6], P[7]); out[2] = make_float4(P[8], P[9], P[10], P[11]);
    }
    // most inliers, then lowest trial id: max over (count << 32 | ~trial); integer atomics are order-independent
    unsigned long long key = (good && cnt > 0) ? (((unsigned long long)(unsigned)cnt << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)trial)) : 0ull;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)key, d, 64), hi = __shfl_xor((unsigned)(key >> 32), d, 64);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        key = other > key ? other : key;
    }
    if ((threadIdx.x & 63) == 0 && key) atomicMax(&best[pair], key);
}

// grid (n_pairs) x 256: ordered inlier list of the winning trial.
__global__ void __launch_bounds__(256) k_ransac_extract(RansacDims D, const float4 *__restrict__ ptsA, const float4 *__restrict__ ptsB, const int *__restrict__ offsets,
                                                       const float *__restrict__ poses, const unsigned long long *__restrict__ best,
                                                       int *__restrict__ inlier_ids, int *__restrict__ n_inliers, int *__restrict__ best_trial, float *__restrict__ best_pose)
{
    __shared__ int wave_cnt[4];
    const int pair = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o = offsets[pair], n = offsets[pair + 1] - o;
    const unsigned long long key = best[pair];
    const int trial = key ? (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) : -1;
    float P[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    if (trial >= 0) {
        const float *src = poses + 12 * ((size_t)pair * D.n_trials + trial);
#pragma unroll
        for (int k = 0; k < 12; k++) P[k] = src[k];
    }
    int base = 0;
    if (trial >= 0)
        for (int i0 = 0; i0 < n; i0 += 256) {
            const int i = i0 + (int)threadIdx.x;
            const bool v = (i < n) && ransac_is_inlier(P, ptsA[o + min(i, n - 1)], ptsB[o + min(i, n - 1)], D.dist_thres);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(v);
            const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (lane == 0) wave_cnt[wave] = __popcll(m);
            __syncthreads();
            int off = base;
            for (int w = 0; w < wave; w++) off += wave_cnt[w];
            if (v) inlier_ids[o + off + before] = i;
            base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            __syncthreads();
        }
    if (threadIdx.x == 0) {
        n_inliers[pair] = base;
        best_trial[pair] = trial;
#pragma unroll
        for (int k = 0; k < 12; k++) best_pose[16 * pair + k] = P[k];
        best_pose[16 * pair + 12] = 0.f; best_pose[16 * pair + 13] = 0.f; best_pose[16 * pair + 14] = 0.f; best_pose[16 * pair + 15] = 1.f;
    }
}

}  // namespace btba
