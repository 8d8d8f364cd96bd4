/* btba_oracle_ransac.c -- CPU restatement of the reference's correspondence RANSAC.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as btba_oracle.c): imported by tests/, never by the product path.
 * PARITY: procrustesKernel and evalPoseKernel of the reference are compiled for the CPU (oracle/Makefile target `ref`,
 * _ref/libbtba_ref_ransac.so) and compared with this file in tests/test_oracle_vs_reference.py.  Beyond that, unpinned:
 *   - the reference holds no golden vectors for this step either;
 *   - two ingredients live in third-party code that is absent from /root/reference and cannot be rebuilt here:
 *     cuRAND's XORWOW generator (curand_init(0, idx, 0) + curand_uniform, cuda_ransac.cu:1156-1163; CUDA toolkit,
 *     version unpinned by the reference) draws the sample triples, and the 3x3 SVD is McAdams et al., "Computing
 *     the SVD of 3x3 matrices with minimal branching and elementary floating point operations" (UW-Madison TR1690,
 *     2011), an APPROXIMATE Jacobi SVD (4 sweeps, rsqrt-based Givens angles) pasted into cuda_ransac.cu:48-975.
 *   What is restated is therefore the algorithm the reference implements around those two: sample triples are an
 *   INPUT (as they are for ransacMultiPairKernel's rand_list, cuda_ransac.cu:1105) or come from the documented
 *   counter hash below, and the rotation is the exact Kabsch optimum the approximate SVD converges to, computed
 *   with a double-precision one-sided Jacobi SVD.
 *
 * Follows: procrustesKernel cuda_ransac.cu:999-1102, evalPoseKernel :978-997, ransacEstimateModelKernel :1145-1181,
 * ransacEvalModelKernel :1183-1200, findBestTrial :1202-1219, ransacMultiPairGPU :1228-1323,
 * SiftManager::runRansacMultiPairGPU FeatureManager.cpp:659-741 (keep inliers; < 5 inliers => drop every match).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* Sample index draw.  Reference: rand_idx = round(curand_uniform(&state) * (n_pts - 1)), u in (0, 1], one XORWOW
 * stream per trial (:1156-1163).  Here u = (h >> 8 + 1) * 2^-24 with h = a 32-bit mix of (seed, pair, trial, draw):
 * same distribution shape (end points get half weight), integer-exact on every platform. */
static uint32_t mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
ORC_API int32_t orc_ransac_draw(uint64_t seed, int32_t pair, int32_t trial, int32_t draw, int32_t n_pts)
{
    uint32_t h = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9E3779B9u * (uint32_t)(pair + 1)));
    h = mix32(h ^ (0x85EBCA6Bu * (uint32_t)(trial + 1)));
    h = mix32(h + 0xC2B2AE35u * (uint32_t)(draw + 1));
    const float u = (float)((h >> 8) + 1u) * (1.0f / 16777216.0f);
    return (int32_t)roundf(u * (float)(n_pts - 1));
}

/* One-sided Jacobi SVD of a 3x3 (double): A = U diag(s) V^T, s descending, U's columns of vanishing singular values
 * completed to a right-handed... (completed orthonormally; handedness is fixed by the caller's determinant rule). */
static void svd3(const double Ain[9], double U[9], double s[3], double V[9])
{
    double A[9], W[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(A, Ain, sizeof A);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < 3; r++) { al += A[3 * r + p] * A[3 * r + p]; be += A[3 * r + q] * A[3 * r + q]; ga += A[3 * r + p] * A[3 * r + q]; }
                if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
                off += fabs(ga);
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int r = 0; r < 3; r++) {
                    const double ap = A[3 * r + p], aq = A[3 * r + q];
                    A[3 * r + p] = c * ap - sn * aq; A[3 * r + q] = sn * ap + c * aq;
                    const double vp = W[3 * r + p], vq = W[3 * r + q];
                    W[3 * r + p] = c * vp - sn * vq; W[3 * r + q] = sn * vp + c * vq;
                }
            }
        if (off == 0) break;
    }
    int order[3] = { 0, 1, 2 };
    double n[3];
    for (int j = 0; j < 3; j++) n[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    for (int a = 0; a < 2; a++) for (int b = a + 1; b < 3; b++) if (n[order[b]] > n[order[a]]) { int t = order[a]; order[a] = order[b]; order[b] = t; }
    for (int j = 0; j < 3; j++) {
        const int o = order[j];
        s[j] = n[o];
        for (int r = 0; r < 3; r++) { V[3 * r + j] = W[3 * r + o]; U[3 * r + j] = n[o] > 0 ? A[3 * r + o] / n[o] : 0.0; }
    }
    /* complete U where a singular value vanished relative to the largest (3-point samples: rank <= 2) */
    const double tiny = 1e-12 * (s[0] > 0 ? s[0] : 1.0);
    if (s[1] <= tiny) {                 /* rank <= 1: any orthonormal completion; callers reject these samples */
        double a[3] = { U[0], U[3], U[6] }, e[3] = { 0, 0, 0 };
        int k = fabs(a[0]) < fabs(a[1]) ? (fabs(a[0]) < fabs(a[2]) ? 0 : 2) : (fabs(a[1]) < fabs(a[2]) ? 1 : 2);
        e[k] = 1;
        if (s[0] <= 0) { a[0] = 1; a[1] = a[2] = 0; U[0] = 1; U[3] = U[6] = 0; e[0] = 0; e[1] = 1; }
        double b[3] = { a[1] * e[2] - a[2] * e[1], a[2] * e[0] - a[0] * e[2], a[0] * e[1] - a[1] * e[0] };
        const double nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
        for (int r = 0; r < 3; r++) U[3 * r + 1] = b[r] / nb;
    }
    if (s[2] <= tiny) {
        const double a[3] = { U[0], U[3], U[6] }, b[3] = { U[1], U[4], U[7] };
        U[2] = a[1] * b[2] - a[2] * b[1]; U[5] = a[2] * b[0] - a[0] * b[2]; U[8] = a[0] * b[1] - a[1] * b[0];
    }
}

static double det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* procrustesKernel (:999-1102).  src/dst: float4 points.  Returns 1 and a row-major 4x4 in pose, or 0 ("R is not
 * valid") with pose = identity.  Means and the 3x3 S are accumulated in fp32 in the reference's order; the rotation
 * R = V U^T (with V's last column flipped when det < 0) is the exact one.  `gap_out` (may be NULL) receives
 * (sigma_2 + d sigma_3) / sigma_1, d = det(V U^T): the margin by which the optimum is unique -- ~0 for collinear samples,
 * where the reference's answer is an artefact of its SVD; callers treat gap < 1e-4 as "not good". */
ORC_API int orc_procrustes(const float *src, const float *dst, int n_pts, float *pose, float *gap_out)
{
    for (int k = 0; k < 16; k++) pose[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    float sm[3] = { 0, 0, 0 }, dm[3] = { 0, 0, 0 };
    for (int i = 0; i < n_pts; i++)
        for (int c = 0; c < 3; c++) { sm[c] += src[4 * i + c]; dm[c] += dst[4 * i + c]; }
    for (int c = 0; c < 3; c++) { sm[c] /= n_pts; dm[c] /= n_pts; }
    float S[9] = { 0 };
    for (int i = 0; i < n_pts; i++) {
        float s[3], d[3];
        for (int c = 0; c < 3; c++) { s[c] = src[4 * i + c] - sm[c]; d[c] = dst[4 * i + c] - dm[c]; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) S[3 * r + c] += s[r] * d[c];
    }
    double Sd[9], U[9], sv[3], V[9], R[9];
    for (int k = 0; k < 9; k++) Sd[k] = S[k];
    svd3(Sd, U, sv, V);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = V[3 * r + 0] * U[3 * c + 0] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2];
    double diff = 0;
    for (int h = 0; h < 3; h++) for (int w = 0; w < 3; w++) {
        double t = -(h == w ? 1.0 : 0.0);
        for (int k = 0; k < 3; k++) t += R[3 * k + h] * R[3 * k + w];
        diff += t * t;
    }
    if (!(sqrt(diff) < 1e-3)) { if (gap_out) *gap_out = 0.0f; return 0; }
    const double d = det3(R) < 0 ? -1.0 : 1.0;
    if (d < 0) {
        for (int r = 0; r < 3; r++) V[3 * r + 2] = -V[3 * r + 2];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = V[3 * r + 0] * U[3 * c + 0] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2];
    }
    if (gap_out) *gap_out = sv[0] > 0 ? (float)((sv[1] + d * sv[2]) / sv[0]) : 0.0f;
    float Rf[9];
    for (int k = 0; k < 9; k++) Rf[k] = (float)R[k];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) pose[4 * r + c] = Rf[3 * r + c];
        pose[4 * r + 3] = dm[r] - (Rf[3 * r] * sm[0] + Rf[3 * r + 1] * sm[1] + Rf[3 * r + 2] * sm[2]);
    }
    return 1;
}

/* inlier test of ransacEvalModelKernel (:1190-1195): pose * ptA (float4x4 * float4, w = 1), Euclidean distance to
 * ptB, REJECT when dist > dist_thres (so dist == dist_thres is an inlier). */
static inline int is_inlier(const float *pose, const float *a, const float *b, float dist_thres)
{
    float d[3];
    for (int r = 0; r < 3; r++) d[r] = b[r] - (pose[4 * r] * a[0] + pose[4 * r + 1] * a[1] + pose[4 * r + 2] * a[2] + pose[4 * r + 3] * a[3]);
    const float dist = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    return !(dist > dist_thres);
}

/* One frame pair of ransacMultiPairGPU (:1228-1323).
 *   samples: n_trials x 3 indices, or NULL to draw them with orc_ransac_draw(seed, pair_id, trial, 0..2, n_pts).
 *   A trial is skipped (0 inliers) when two of its indices coincide or one is negative (:1164-1165), when
 *   procrustes fails, or when the sample is (near-)collinear (gap < 1e-4, see orc_procrustes).
 *   Best trial: most inliers; among equals the LOWEST trial id (the reference's findBestTrial lets whichever thread
 *   writes last win -- any of the maxima; a fixed rule is needed to compare two implementations).
 *   Outputs: inlier_ids (ascending, capacity n_pts), *n_inliers, *best_trial (-1 if no trial was good), best_pose[16],
 *   and optionally counts[n_trials] (inliers per trial; 0 for skipped trials), poses_out[n_trials*16]. */
ORC_API int orc_ransac_pair(const float *ptsA, const float *ptsB, int n_pts, int n_trials, float dist_thres,
                            const int32_t *samples, uint64_t seed, int pair_id,
                            int32_t *inlier_ids, int32_t *n_inliers, int32_t *best_trial, float *best_pose,
                            int32_t *counts, float *poses_out)
{
    int best = -1, best_cnt = 0;
    float bp[16];
    for (int k = 0; k < 16; k++) bp[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    for (int t = 0; t < n_trials; t++) {
        int idx[3];
        for (int k = 0; k < 3; k++) idx[k] = samples ? samples[3 * t + k] : orc_ransac_draw(seed, pair_id, t, k, n_pts);
        float pose[16];
        for (int k = 0; k < 16; k++) pose[k] = (k % 5 == 0) ? 1.0f : 0.0f;
        int cnt = 0, good = 0;
        if (n_pts >= 3 && !(idx[0] == idx[1] || idx[1] == idx[2] || idx[0] == idx[2]) && !(idx[0] < 0 || idx[1] < 0 || idx[2] < 0) &&
            idx[0] < n_pts && idx[1] < n_pts && idx[2] < n_pts) {
            float s[12], d[12], gap;
            for (int k = 0; k < 3; k++) { memcpy(s + 4 * k, ptsA + 4 * idx[k], 16); memcpy(d + 4 * k, ptsB + 4 * idx[k], 16); }
            good = orc_procrustes(s, d, 3, pose, &gap) && gap >= 1e-4f;
        }
        if (good)
            for (int i = 0; i < n_pts; i++) cnt += is_inlier(pose, ptsA + 4 * i, ptsB + 4 * i, dist_thres);
        if (counts) counts[t] = good ? cnt : 0;
        if (poses_out) memcpy(poses_out + 16 * t, pose, sizeof pose);
        if (good && cnt > best_cnt) { best_cnt = cnt; best = t; memcpy(bp, pose, sizeof bp); }
    }
    int n = 0;
    if (best >= 0)
        for (int i = 0; i < n_pts; i++) if (is_inlier(bp, ptsA + 4 * i, ptsB + 4 * i, dist_thres)) inlier_ids[n++] = i;
    *n_inliers = n;
    *best_trial = best;
    if (best_pose) memcpy(best_pose, bp, sizeof bp);
    return 0;
}
