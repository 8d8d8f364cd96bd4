/* btba_oracle_ransac.c -- CPU restatement of the reference's correspondence RANSAC.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as btba_oracle.c): imported by tests/, never by the product path.
 * PARITY: procrustesKernel and evalPoseKernel of the reference are compiled for the CPU (oracle/Makefile target `ref`,
 * _ref/libbtba_ref_ransac.so) and compared with this file in tests/test_oracle_vs_reference.py.  Beyond that, unpinned:
 *   - the reference holds no golden vectors for this step either;
 *   - the sample triples come from cuRAND's XORWOW generator (curand_init(0, idx, 0) + curand_uniform,
 *     cuda_ransac.cu:1154-1161) -- CUDA toolkit code, absent from /root/reference and from this image.  Its published
 *     algorithm is restated in oracle/xorwow.h (see that header for what each part is anchored on: the recurrence and the
 *     2^67 subsequence jump are checked against rocRAND's tables, the four seed-scrambling constants are unpinned) and
 *     exported below (orc_ransac_reference_samples); sample triples can also be an INPUT (as they are for
 *     ransacMultiPairKernel's rand_list, cuda_ransac.cu:1105) or come from the documented counter hash below;
 *   - the 3x3 SVD is McAdams et al., "Computing the SVD of 3x3 matrices with minimal branching and elementary floating
 *     point operations" (UW-Madison TR1690, 2011), an APPROXIMATE Jacobi SVD (4 sweeps, rsqrt-based Givens angles) pasted
 *     into cuda_ransac.cu:48-975: restated operation for operation (mc_svd, orc_procrustes_reference: hypothesis 0) and
 *     pinned bit for bit against the reference's own code; hypothesis 1 is the exact Kabsch optimum the approximate SVD
 *     converges to, computed with a double-precision one-sided Jacobi SVD.
 *   The reference's kernels and host launcher themselves (ransacEstimateModelKernel .. ransacMultiPairGPU) are compiled for
 *   the CPU as well (oracle/Makefile, _ref/libbtba_ref_ransac.so: ref_ransac_multi_pair) with oracle/xorwow.h standing in
 *   for <curand_kernel.h>, and compared end to end with orc_ransac_pair_ex on orc_ransac_reference_samples.
 *
 * Follows: procrustesKernel cuda_ransac.cu:999-1102, evalPoseKernel :978-997, ransacEstimateModelKernel :1145-1181,
 * ransacEvalModelKernel :1183-1200, findBestTrial :1202-1219, ransacMultiPairGPU :1228-1323,
 * SiftManager::runRansacMultiPairGPU FeatureManager.cpp:659-741 (keep inliers; < 5 inliers => drop every match).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "xorwow.h"

#define ORC_API __attribute__((visibility("default")))

/* ---- the reference's sample stream (cuda_ransac.cu:1154-1161): trial t draws three uniforms from its own XORWOW stream
 * curand_init(seed = 0, subsequence = t, offset = 0) and turns each into round(u * (n_pts - 1)).  The stream does not
 * depend on the frame pair, so the n_trials x 3 uniforms are one table for all pairs. */
ORC_API void orc_curand_xorwow_state(uint64_t seed, uint64_t subsequence, uint64_t offset, uint32_t state_out[6] /* d, v[0..4] */)
{
    orc_xorwow_state s;
    orc_curand_init(seed, subsequence, offset, &s);
    state_out[0] = s.d;
    memcpy(state_out + 1, s.v, sizeof s.v);
}
ORC_API void orc_curand_xorwow_draw(uint64_t seed, uint64_t subsequence, uint64_t offset, int n, uint32_t *raw_out /* may be NULL */, float *uniform_out /* may be NULL */)
{
    orc_xorwow_state s;
    orc_curand_init(seed, subsequence, offset, &s);
    for (int i = 0; i < n; i++) {
        orc_xorwow_state c = s;
        const uint32_t x = orc_xorwow_next(&s);
        if (raw_out) raw_out[i] = x;
        if (uniform_out) uniform_out[i] = orc_curand_uniform(&c);
    }
}
ORC_API void orc_ransac_reference_uniforms(uint64_t seed, int n_trials, float *u_out /* [n_trials][3] */)
{
    for (int t = 0; t < n_trials; t++) {
        orc_xorwow_state s;
        orc_curand_init(seed, (uint64_t)t, 0, &s);
        for (int k = 0; k < 3; k++) u_out[3 * t + k] = orc_curand_uniform(&s);
    }
}
ORC_API void orc_ransac_reference_samples(uint64_t seed, int n_trials, int n_pts, int32_t *samples_out /* [n_trials][3] */)
{
    for (int t = 0; t < n_trials; t++) {
        orc_xorwow_state s;
        orc_curand_init(seed, (uint64_t)t, 0, &s);
        for (int k = 0; k < 3; k++) samples_out[3 * t + k] = (int32_t)roundf(orc_curand_uniform(&s) * (float)(n_pts - 1));   /* :1159-1161 */
    }
}
/* A^(2^k) (which = 0) or A^(2^(67 + k)) (which = 1), [160 input bits][5 output words] -- rocRAND's table layout */
ORC_API void orc_xorwow_matrix(int which, int k, uint32_t *out /* 800 */)
{
    memcpy(out, orc_xorwow_power(which, k)->col, 800 * sizeof(uint32_t));
}

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

/* ---- the reference's own hypothesis: procrustesKernel with its APPROXIMATE 3x3 SVD ------------------------------------
 * McAdams, Selle, Tamstorf, Teran, Sifakis: "Computing the Singular Value Decomposition of 3x3 matrices with minimal branching
 * and elementary floating point operations", UW-Madison TR1690 (2011); the reference carries the report's scalar code,
 * macro-expanded, in cuda_ransac.cu:48-975.  Restated here stage by stage in the report's operation order (every sum one IEEE
 * fp32 operation, no contraction; rsqrt correctly rounded like __frsqrt_rn), pinned BIT FOR BIT against that code compiled for
 * the CPU (tests/test_oracle_vs_reference.py).  Four sweeps of three approximate Jacobi conjugations, V from the accumulated
 * quaternion, B = A V, columns sorted by norm, Givens QR. */
typedef struct { float s11, s21, s31, s22, s32, s33, qs, qx, qy, qz; } mc_state;
/* correctly rounded x^-1/2 (__frsqrt_rn): the double result rounded to float, settled exactly at the two neighbouring midpoints by the sign
 * of fma(m^2, x, -1) (m^2 is exact in double; one rounding) -- see bundletrack_amd/csrc/btba_svd3.hpp::rsqrt_rn, written separately */
static float mc_rsqrt(float x)
{
    float r = (float)(1.0 / sqrt((double)x));
    uint32_t rb; memcpy(&rb, &r, 4);
    if (!(r > 0.0f) || (rb & 0x7F800000u) == 0x7F800000u || (rb & 0x7F800000u) == 0u) return r;
    uint32_t ub = rb + 1u, db = rb - 1u; float up, dn; memcpy(&up, &ub, 4); memcpy(&dn, &db, 4);
    const double mh = 0.5 * ((double)r + (double)up), ml = 0.5 * ((double)r + (double)dn);
    if (fma(mh * mh, (double)x, -1.0) < 0.0) return up;
    if (fma(ml * ml, (double)x, -1.0) > 0.0) return dn;
    return r;
}
static float mc_rsqrt1(float x)                /* one Newton step: r + r/2 - x r (r (r/2)) */
{
    const float r = mc_rsqrt(x), h = r * 0.5f;
    volatile float t = r * h; t = r * t; t = x * t;
    volatile float u = r + h; u = u - t;
    return u;
}
static float mc_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
/* conjugation in the (p, q) plane: pp, qp, qq the 2x2 block; rp, rq the remaining row; rr the remaining diagonal; (a, b, c) the
 * quaternion's vector part rotated to (p, q, r) */
static void mc_conj(float *pp, float *qp, float *rp, float *qq, float *rq, float *rr, float *qs, float *a, float *b, float *c)
{
    volatile float sh = *qp * 0.5f, d = *pp - *qq, t1, t2, t3, t4, ch, cs, sn;
    t2 = sh * sh;
    if (!(t2 >= 1.e-20f)) { sh = 0.0f; ch = 1.0f; } else ch = d;
    t1 = sh * sh; t2 = ch * ch; t3 = t1 + t2; t4 = mc_rsqrt(t3);
    sh = t4 * sh; ch = t4 * ch;
    t1 = 5.8284273147583007813f * t1;
    if (t2 <= t1) { sh = mc_bits(1053028117u); ch = mc_bits(1064076127u); }        /* sin, cos of pi / 8 */
    t1 = sh * sh; t2 = ch * ch; cs = t2 - t1; sn = ch * sh; sn = sn + sn;
    t3 = t1 + t2;
    *rr = *rr * t3; *rp = *rp * t3; *rq = *rq * t3; *rr = *rr * t3;
    t1 = sn * *rp; t2 = sn * *rq; *rp = cs * *rp; *rq = cs * *rq; *rp = t2 + *rp; *rq = *rq - t1;
    t2 = sn * sn; t1 = *qq * t2; t3 = *pp * t2; t4 = cs * cs;
    *pp = *pp * t4; *qq = *qq * t4; *pp = *pp + t1; *qq = *qq + t3;
    t4 = t4 - t2; t2 = *qp + *qp; *qp = *qp * t4; t4 = cs * sn; t2 = t2 * t4; d = d * t4;
    *pp = *pp + t2; *qp = *qp - d; *qq = *qq - t2;
    t1 = sh * *a; t2 = sh * *b; t3 = sh * *c; sh = sh * *qs;
    *qs = ch * *qs; *a = ch * *a; *b = ch * *b; *c = ch * *c;
    *c = *c + sh; *qs = *qs - t3; *a = *a + t2; *b = *b - t1;
}
static void mc_sort(float *B, float *V, float *na, float *nb, int ca, int cb, int neg)
{
    const int sw = *na < *nb;
    for (int r = 0; r < 3; r++) {
        if (sw) { float t = B[3 * r + ca]; B[3 * r + ca] = B[3 * r + cb]; B[3 * r + cb] = t; t = V[3 * r + ca]; V[3 * r + ca] = V[3 * r + cb]; V[3 * r + cb] = t; }
    }
    if (sw) { const float t = *na; *na = *nb; *nb = t; }
    const float sg = 1.0f + (sw ? -2.0f : 0.0f);
    for (int r = 0; r < 3; r++) { B[3 * r + neg] *= sg; V[3 * r + neg] *= sg; }
}
static void mc_givens(float *B, float *U, int p, int q)
{
    const float app = B[3 * p + p], aqp = B[3 * q + p];
    volatile float sh = aqp * aqp, ch, t1, t2, cs, sn;
    sh = (sh >= 1.e-12f) ? aqp : 0.0f;
    ch = 0.0f - app; ch = fmaxf(ch, app); ch = fmaxf(ch, 1.e-12f);
    t1 = ch * ch; t2 = sh * sh; t2 = t1 + t2; t1 = mc_rsqrt1(t2); t1 = t1 * t2;
    ch = ch + t1;
    if (!(app >= 0.0f)) { const float t = ch; ch = sh; sh = t; }
    t1 = ch * ch; t2 = sh * sh; t2 = t1 + t2; t1 = mc_rsqrt1(t2);
    ch = ch * t1; sh = sh * t1;
    cs = ch * ch; sn = sh * sh; cs = cs - sn; sn = sh * ch; sn = sn + sn;
    for (int k = 0; k < 3; k++) {
        volatile float x = B[3 * p + k], y = B[3 * q + k], sx = sn * x, sy = sn * y, cx = cs * x, cy = cs * y;
        B[3 * p + k] = cx + sy; B[3 * q + k] = cy - sx;
        x = U[3 * k + p]; y = U[3 * k + q]; sx = sn * x; sy = sn * y; cx = cs * x; cy = cs * y;
        U[3 * k + p] = cx + sy; U[3 * k + q] = cy - sx;
    }
}
static void mc_svd(const float *A, float *U, float *sig, float *V)
{
    mc_state m;
    float *S[6] = { &m.s11, &m.s21, &m.s31, &m.s22, &m.s32, &m.s33 };
    static const int ca[6] = { 0, 1, 2, 1, 2, 2 }, cb[6] = { 0, 0, 0, 1, 1, 2 };
    for (int k = 0; k < 6; k++) {                   /* S = A^T A, summed top to bottom */
        volatile float t = A[ca[k]] * A[cb[k]], u = A[3 + ca[k]] * A[3 + cb[k]];
        t = u + t; u = A[6 + ca[k]] * A[6 + cb[k]]; t = u + t;
        *S[k] = t;
    }
    m.qs = 1.0f; m.qx = m.qy = m.qz = 0.0f;
    for (int sweep = 0; sweep < 4; sweep++) {
        mc_conj(&m.s11, &m.s21, &m.s31, &m.s22, &m.s32, &m.s33, &m.qs, &m.qx, &m.qy, &m.qz);
        mc_conj(&m.s22, &m.s32, &m.s21, &m.s33, &m.s31, &m.s11, &m.qs, &m.qy, &m.qz, &m.qx);
        mc_conj(&m.s33, &m.s31, &m.s32, &m.s11, &m.s21, &m.s22, &m.qs, &m.qz, &m.qx, &m.qy);
    }
    volatile float t1, t2, t3;
    t2 = m.qs * m.qs; t1 = m.qx * m.qx; t2 = t1 + t2; t1 = m.qy * m.qy; t2 = t1 + t2; t1 = m.qz * m.qz; t2 = t1 + t2;
    const float nr = mc_rsqrt1(t2);
    const float qs = m.qs * nr, qx = m.qx * nr, qy = m.qy * nr, qz = m.qz * nr;
    volatile float v11, v22, v33, v12, v13, v21, v23, v31, v32;
    t1 = qx * qx; t2 = qy * qy; t3 = qz * qz;
    v11 = qs * qs; v22 = v11 - t1; v33 = v22 - t2; v33 = v33 + t3; v22 = v22 + t2; v22 = v22 - t3; v11 = v11 + t1; v11 = v11 - t2; v11 = v11 - t3;
    t1 = qx + qx; t2 = qy + qy; t3 = qz + qz;
    v32 = qs * t1; v13 = qs * t2; v21 = qs * t3;
    t1 = qy * t1; t2 = qz * t2; t3 = qx * t3;
    v12 = t1 - v21; v23 = t2 - v32; v31 = t3 - v13; v21 = t1 + v21; v32 = t2 + v32; v13 = t3 + v13;
    V[0] = v11; V[1] = v12; V[2] = v13; V[3] = v21; V[4] = v22; V[5] = v23; V[6] = v31; V[7] = v32; V[8] = v33;
    float B[9];
    for (int r = 0; r < 3; r++) {                   /* B = A V */
        const float a1 = A[3 * r], a2 = A[3 * r + 1], a3 = A[3 * r + 2];
        volatile float b1 = v11 * a1, b2 = v12 * a1, b3 = v13 * a1, t;
        t = v21 * a2; b1 = b1 + t; t = v31 * a3; b1 = b1 + t;
        t = v22 * a2; b2 = b2 + t; t = v32 * a3; b2 = b2 + t;
        t = v23 * a2; b3 = b3 + t; t = v33 * a3; b3 = b3 + t;
        B[3 * r] = b1; B[3 * r + 1] = b2; B[3 * r + 2] = b3;
    }
    float n[3];
    for (int c = 0; c < 3; c++) { volatile float t = B[c] * B[c], u = B[3 + c] * B[3 + c]; t = t + u; u = B[6 + c] * B[6 + c]; t = t + u; n[c] = t; }
    mc_sort(B, V, &n[0], &n[1], 0, 1, 1);
    mc_sort(B, V, &n[0], &n[2], 0, 2, 0);
    mc_sort(B, V, &n[1], &n[2], 1, 2, 2);
    for (int k = 0; k < 9; k++) U[k] = (k % 4 == 0) ? 1.0f : 0.0f;
    mc_givens(B, U, 0, 1); mc_givens(B, U, 0, 2); mc_givens(B, U, 1, 2);
    sig[0] = B[0]; sig[1] = B[4]; sig[2] = B[8];
}

/* procrustesKernel (:998-1103) as the reference runs it: fp32 means and correlation, the SVD above, R = V U^T, "R is not valid"
 * when |R^T R - I|_F >= 1e-3 (returns 0, pose = identity), V's last column flipped when det R < 0, t = dst_mean - R src_mean. */
ORC_API int orc_procrustes_reference(const float *src, const float *dst, int n_pts, float *pose)
{
    for (int k = 0; k < 16; k++) pose[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    float sm[3] = { 0, 0, 0 }, dm[3] = { 0, 0, 0 };
    for (int i = 0; i < n_pts; i++)
        for (int c = 0; c < 3; c++) { sm[c] += src[4 * i + c]; dm[c] += dst[4 * i + c]; }
    for (int c = 0; c < 3; c++) { sm[c] /= (float)n_pts; dm[c] /= (float)n_pts; }
    float S[9] = { 0 };
    for (int i = 0; i < n_pts; i++) {
        float s[3], d[3];
        for (int c = 0; c < 3; c++) { s[c] = src[4 * i + c] - sm[c]; d[c] = dst[4 * i + c] - dm[c]; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) S[3 * r + c] += s[r] * d[c];
    }
    float U[9], V[9], sig[3], R[9];
    mc_svd(S, U, sig, V);
    for (int pass = 0; pass < 2; pass++) {
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2];
        if (pass == 1) break;
        float diff = 0.0f;
        for (int h = 0; h < 3; h++) for (int w = 0; w < 3; w++) {
            const float t = (R[h] * R[w] + R[3 + h] * R[3 + w] + R[6 + h] * R[6 + w]) - (h == w ? 1.0f : 0.0f);
            diff += t * t;
        }
        diff = sqrtf(diff);
        if ((double)diff >= 1e-3) return 0;
        const float det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[6] * R[4] * R[2] - R[7] * R[5] * R[0] - R[8] * R[3] * R[1];
        if (!(det < 0.0f)) break;
        for (int r = 0; r < 3; r++) V[3 * r + 2] = -V[3 * r + 2];
    }
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) pose[4 * r + c] = R[3 * r + c];
        pose[4 * r + 3] = dm[r] - (R[3 * r] * sm[0] + R[3 * r + 1] * sm[1] + R[3 * r + 2] * sm[2]);
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
ORC_API int orc_ransac_pair_ex(int hypothesis /* 0: the reference's approximate-SVD procrustes; 1: exact Kabsch + collinearity gap */,
                               const float *ptsA, const float *ptsB, int n_pts, int n_trials, float dist_thres,
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
            good = hypothesis == 0 ? orc_procrustes_reference(s, d, 3, pose) : (orc_procrustes(s, d, 3, pose, &gap) && gap >= 1e-4f);
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

ORC_API int orc_ransac_pair(const float *ptsA, const float *ptsB, int n_pts, int n_trials, float dist_thres,
                            const int32_t *samples, uint64_t seed, int pair_id,
                            int32_t *inlier_ids, int32_t *n_inliers, int32_t *best_trial, float *best_pose,
                            int32_t *counts, float *poses_out)
{
    return orc_ransac_pair_ex(1, ptsA, ptsB, n_pts, n_trials, dist_thres, samples, seed, pair_id, inlier_ids, n_inliers, best_trial, best_pose, counts, poses_out);
}
