// btba_svd3.hpp -- the 3x3 SVD the reference's RANSAC uses, restated: McAdams, Selle, Tamstorf, Teran, Sifakis,
// "Computing the Singular Value Decomposition of 3x3 matrices with minimal branching and elementary floating point
// operations", UW-Madison TR1690 (2011).  The reference carries the authors' scalar code, macro-expanded, in
// src/cuda/cuda_ransac.cu:48-975 and calls it from procrustesKernel (:998-1103).  It is an APPROXIMATE SVD -- four
// fixed sweeps of Jacobi conjugations with rsqrt-based approximate Givens angles, then a Givens QR -- so a hypothesis
// built on it differs from the exact Kabsch optimum (on 3-point samples by more than 4e-3 in ~5 % of the cases);
// reproducing the reference's per-trial poses, inlier counts and winner therefore needs THIS arithmetic, operation for
// operation.  The three stages below are the report's kernels written once as functions of their operand roles and
// instantiated with the report's index permutations; every sum is a single IEEE operation in the report's order
// (no contraction), the reciprocal square root is correctly rounded like CUDA's __frsqrt_rn.
//
// Plain C++ (host and device): tests/cpp compiles it with g++ and holds it bit for bit against the reference's own
// function (oracle/_ref/libbtba_ref_ransac.so).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define BTBA_HD __host__ __device__ __forceinline__
#else
#define BTBA_HD static inline
#endif

namespace btba {
namespace svd3 {

BTBA_HD float f_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
BTBA_HD uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
// __frsqrt_rn: the CORRECTLY ROUNDED reciprocal square root.  y = 1 / sqrt(x) in double carries two roundings and the cast a third, so
// (float) y can land on the wrong side when y falls within an ulp(double) of the midpoint of two floats (about one input in 2^29).  Those
// are settled exactly: at a midpoint m (25 significant bits, m^2 exact in double) the sign of fma(m^2, x, -1) -- ONE rounding of the exact
// m^2 x - 1 -- says on which side of m the true x^-1/2 lies (it is never ON a midpoint: m^2 x = 1 would make m a power of two).
BTBA_HD float rsqrt_rn(float x)
{
    float r = (float)(1.0 / sqrt((double)x));
    const uint32_t rb = bits(r);
    if (!(r > 0.0f) || (rb & 0x7F800000u) == 0x7F800000u || (rb & 0x7F800000u) == 0u) return r;      // 0, inf, NaN, subnormal results: as computed
    const float up = f_from_bits(rb + 1u), dn = f_from_bits(rb - 1u);
    const double xd = (double)x, m_hi = 0.5 * ((double)r + (double)up), m_lo = 0.5 * ((double)r + (double)dn);
    if (fma(m_hi * m_hi, xd, -1.0) < 0.0) r = up;               // x^-1/2 above the upper midpoint
    else if (fma(m_lo * m_lo, xd, -1.0) > 0.0) r = dn;          // ... below the lower one
    return r;
}
// one Newton step on the reciprocal square root, in the report's operation order: r (1.5 - 0.5 x r^2) as r + r/2 - x r (r (r/2))
BTBA_HD float rsqrt_refined(float x)
{
#pragma clang fp contract(off)
    const float r = rsqrt_rn(x);
    const float h = r * 0.5f;
    float t = r * h;
    t = r * t;
    t = x * t;
    return (r + h) - t;
}

constexpr float kFourGammaSquared = 5.8284273147583007813f;      // 3 + 2 sqrt 2
constexpr uint32_t kSinPiOver8 = 1053028117u, kCosPiOver8 = 1064076127u;       // bit patterns, as in the report
constexpr float kTiny = 1.e-20f, kSmall = 1.e-12f;

// Jacobi conjugation of the symmetric S = A^T A by an approximate Givens rotation in the (p, q) plane, accumulated into the
// quaternion (qs, q[3]).  Roles: spp, sqp, sqq the 2x2 block that is diagonalised; srp, srq the remaining row; srr the
// remaining diagonal entry; (qa, qb, qc) the quaternion's vector part rotated to (p, q, r).
BTBA_HD void jacobi_conjugation(float &spp, float &sqp, float &srp, float &sqq, float &srq, float &srr, float &qs, float &qa, float &qb, float &qc)
{
#pragma clang fp contract(off)
    float sh = sqp * 0.5f;
    float d = spp - sqq;
    float t2 = sh * sh;
    const bool big = t2 >= kTiny;
    sh = big ? sh : 0.0f;
    float ch = big ? d : 1.0f;
    float t1 = sh * sh;
    t2 = ch * ch;
    float t3 = t1 + t2;
    const float t4 = rsqrt_rn(t3);
    sh = t4 * sh;
    ch = t4 * ch;
    t1 = kFourGammaSquared * t1;
    const bool use_pi8 = t2 <= t1;                      // the rotation would exceed pi/4: take pi/8 instead
    sh = use_pi8 ? f_from_bits(kSinPiOver8) : sh;
    ch = use_pi8 ? f_from_bits(kCosPiOver8) : ch;
    t1 = sh * sh;
    t2 = ch * ch;
    const float c = t2 - t1;
    float s = ch * sh;
    s = s + s;
    // conjugate S (the factor sh^2 + ch^2 keeps the scale the approximate angle would lose)
    t3 = t1 + t2;
    srr = srr * t3;
    srp = srp * t3;
    srq = srq * t3;
    srr = srr * t3;
    t1 = s * srp;
    t2 = s * srq;
    srp = c * srp;
    srq = c * srq;
    srp = t2 + srp;
    srq = srq - t1;
    t2 = s * s;
    t1 = sqq * t2;
    t3 = spp * t2;
    float t4b = c * c;
    spp = spp * t4b;
    sqq = sqq * t4b;
    spp = spp + t1;
    sqq = sqq + t3;
    t4b = t4b - t2;
    t2 = sqp + sqp;
    sqp = sqp * t4b;
    t4b = c * s;
    t2 = t2 * t4b;
    d = d * t4b;
    spp = spp + t2;
    sqp = sqp - d;
    sqq = sqq - t2;
    // accumulate the rotation into the quaternion
    t1 = sh * qa;
    t2 = sh * qb;
    t3 = sh * qc;
    sh = sh * qs;
    qs = ch * qs;
    qa = ch * qa;
    qb = ch * qb;
    qc = ch * qc;
    qc = qc + sh;
    qs = qs - t3;
    qa = qa + t2;
    qb = qb - t1;
}

// swap columns (a, b) of B and V when column a is shorter, then flip the sign of column `neg` (keeps det V = +1)
BTBA_HD void cond_swap(bool c, float &x, float &y) { const float tx = x, ty = y; x = c ? ty : tx; y = c ? tx : ty; }
BTBA_HD void sort_columns(float (&B)[9], float (&V)[9], float &na, float &nb, int ca, int cb, int neg)
{
#pragma clang fp contract(off)
    const bool sw = na < nb;
    for (int r = 0; r < 3; r++) { cond_swap(sw, B[3 * r + ca], B[3 * r + cb]); cond_swap(sw, V[3 * r + ca], V[3 * r + cb]); }
    cond_swap(sw, na, nb);
    const float sgn = 1.0f + (sw ? -2.0f : 0.0f);
    for (int r = 0; r < 3; r++) { B[3 * r + neg] = B[3 * r + neg] * sgn; V[3 * r + neg] = V[3 * r + neg] * sgn; }
}

// Givens rotation of rows (p, q) of B that annihilates B[q][pivot column = p], applied to U's columns (p, q)
BTBA_HD void qr_givens(float (&B)[9], float (&U)[9], int p, int q)
{
#pragma clang fp contract(off)
    const float app = B[3 * p + p], aqp = B[3 * q + p];
    float sh = aqp * aqp;
    sh = (sh >= kSmall) ? aqp : 0.0f;
    float ch = 0.0f - app;
    ch = fmaxf(ch, app);
    ch = fmaxf(ch, kSmall);
    const bool pos = app >= 0.0f;
    float t1 = ch * ch;
    float t2 = sh * sh;
    t2 = t1 + t2;
    t1 = rsqrt_refined(t2);
    t1 = t1 * t2;                                      // = sqrt(ch^2 + sh^2)
    ch = ch + t1;
    { const float a = ch, b = sh; ch = pos ? a : b; sh = pos ? b : a; }
    t1 = ch * ch;
    t2 = sh * sh;
    t2 = t1 + t2;
    t1 = rsqrt_refined(t2);
    ch = ch * t1;
    sh = sh * t1;
    float c = ch * ch;
    float s = sh * sh;
    c = c - s;
    s = sh * ch;
    s = s + s;
    for (int k = 0; k < 3; k++) {                      // rows p, q of B
        const float x = B[3 * p + k], y = B[3 * q + k];
        const float sx = s * x, sy = s * y;
        B[3 * p + k] = c * x + sy;
        B[3 * q + k] = c * y - sx;
    }
    for (int k = 0; k < 3; k++) {                      // columns p, q of U
        const float x = U[3 * k + p], y = U[3 * k + q];
        const float sx = s * x, sy = s * y;
        U[3 * k + p] = c * x + sy;
        U[3 * k + q] = c * y - sx;
    }
}

// A (row-major 3x3) ~= U diag(sig) V^T
BTBA_HD void svd(const float (&A)[9], float (&U)[9], float (&sig)[3], float (&V)[9])
{
#pragma clang fp contract(off)
    // S = A^T A (lower triangle), column dot products summed top to bottom
    auto col_dot = [&](int a, int b) { float s = A[a] * A[b]; s = A[3 + a] * A[3 + b] + s; s = A[6 + a] * A[6 + b] + s; return s; };
    float s11 = col_dot(0, 0), s21 = col_dot(1, 0), s31 = col_dot(2, 0), s22 = col_dot(1, 1), s32 = col_dot(2, 1), s33 = col_dot(2, 2);
    float qs = 1.0f, qx = 0.0f, qy = 0.0f, qz = 0.0f;
    for (int sweep = 0; sweep < 4; sweep++) {
        jacobi_conjugation(s11, s21, s31, s22, s32, s33, qs, qx, qy, qz);
        jacobi_conjugation(s22, s32, s21, s33, s31, s11, qs, qy, qz, qx);
        jacobi_conjugation(s33, s31, s32, s11, s21, s22, qs, qz, qx, qy);
    }
    // normalise the quaternion (one Newton step on the correctly rounded rsqrt), then V from it
    float t2 = qs * qs;
    t2 = qx * qx + t2;
    t2 = qy * qy + t2;
    t2 = qz * qz + t2;
    const float nrm = rsqrt_refined(t2);
    qs = qs * nrm; qx = qx * nrm; qy = qy * nrm; qz = qz * nrm;
    float t1 = qx * qx;
    t2 = qy * qy;
    float t3 = qz * qz;
    float v11 = qs * qs;
    float v22 = v11 - t1;
    float v33 = v22 - t2;
    v33 = v33 + t3;
    v22 = v22 + t2;
    v22 = v22 - t3;
    v11 = v11 + t1;
    v11 = v11 - t2;
    v11 = v11 - t3;
    t1 = qx + qx;
    t2 = qy + qy;
    t3 = qz + qz;
    float v32 = qs * t1, v13 = qs * t2, v21 = qs * t3;
    t1 = qy * t1;
    t2 = qz * t2;
    t3 = qx * t3;
    const float v12 = t1 - v21, v23 = t2 - v32, v31 = t3 - v13;
    v21 = t1 + v21;
    v32 = t2 + v32;
    v13 = t3 + v13;
    V[0] = v11; V[1] = v12; V[2] = v13; V[3] = v21; V[4] = v22; V[5] = v23; V[6] = v31; V[7] = v32; V[8] = v33;
    // B = A V, row by row
    float B[9];
    for (int r = 0; r < 3; r++) {
        const float a1 = A[3 * r], a2 = A[3 * r + 1], a3 = A[3 * r + 2];
        float b1 = v11 * a1, b2 = v12 * a1, b3 = v13 * a1;
        b1 = b1 + v21 * a2;
        b1 = b1 + v31 * a3;
        b2 = b2 + v22 * a2;
        b2 = b2 + v32 * a3;
        b3 = b3 + v23 * a2;
        b3 = b3 + v33 * a3;
        B[3 * r] = b1; B[3 * r + 1] = b2; B[3 * r + 2] = b3;
    }
    // squared column norms, then the three conditional swaps that sort them descending
    auto col_norm2 = [&](int c) { float s = B[c] * B[c]; s = s + B[3 + c] * B[3 + c]; s = s + B[6 + c] * B[6 + c]; return s; };
    float n1 = col_norm2(0), n2 = col_norm2(1), n3 = col_norm2(2);
    sort_columns(B, V, n1, n2, 0, 1, 1);
    sort_columns(B, V, n1, n3, 0, 2, 0);
    sort_columns(B, V, n2, n3, 1, 2, 2);
    // QR by three Givens rotations: B = U R, R's diagonal = the singular values
    for (int k = 0; k < 9; k++) U[k] = (k % 4 == 0) ? 1.0f : 0.0f;
    qr_givens(B, U, 0, 1);
    qr_givens(B, U, 0, 2);
    qr_givens(B, U, 1, 2);
    sig[0] = B[0]; sig[1] = B[4]; sig[2] = B[8];
}

// procrustesKernel (cuda_ransac.cu:998-1103) on n points (xyz of float4): means, correlation S, SVD, R = V U^T, the
// "R is not valid" test (|R^T R - I|_F >= 1e-3: returns false, pose = identity), the reflection fix (det R < 0: flip V's
// last column), t = dst_mean - R src_mean.  P = 3x4 row-major.
template <class PtS, class PtD>
BTBA_HD bool procrustes_reference(const PtS *src, const PtD *dst, int n, float (&P)[12])
{
#pragma clang fp contract(off)
    for (int k = 0; k < 12; k++) P[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    float sm[3] = { 0, 0, 0 }, dm[3] = { 0, 0, 0 };
    for (int i = 0; i < n; i++) {
        sm[0] += src[i].x; sm[1] += src[i].y; sm[2] += src[i].z;
        dm[0] += dst[i].x; dm[1] += dst[i].y; dm[2] += dst[i].z;
    }
    for (int c = 0; c < 3; c++) { sm[c] /= (float)n; dm[c] /= (float)n; }
    float S[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < n; i++) {
        const float s[3] = { src[i].x - sm[0], src[i].y - sm[1], src[i].z - sm[2] }, d[3] = { dst[i].x - dm[0], dst[i].y - dm[1], dst[i].z - dm[2] };
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) S[3 * r + c] += s[r] * d[c];
    }
    float U[9], V[9], sig[3];
    svd(S, U, sig, V);
    auto v_ut = [&](float (&R)[9]) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2]; };
    float R[9];
    v_ut(R);
    float diff = 0.0f;
    for (int h = 0; h < 3; h++)
        for (int w = 0; w < 3; w++) {
            const float t = (R[h] * R[w] + R[3 + h] * R[3 + w] + R[6 + h] * R[6 + w]) - (h == w ? 1.0f : 0.0f);       // (R^T R - I)(h, w)
            diff += t * t;
        }
    diff = sqrtf(diff);
    if ((double)diff >= 1e-3) return false;                                   // "R is not valid"
    const float det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[6] * R[4] * R[2] - R[7] * R[5] * R[0] - R[8] * R[3] * R[1];
    if (det < 0.0f) {
        for (int r = 0; r < 3; r++) V[3 * r + 2] = -V[3 * r + 2];
        v_ut(R);
    }
    for (int r = 0; r < 3; r++) {
        P[4 * r] = R[3 * r]; P[4 * r + 1] = R[3 * r + 1]; P[4 * r + 2] = R[3 * r + 2];
        P[4 * r + 3] = dm[r] - (R[3 * r] * sm[0] + R[3 * r + 1] * sm[1] + R[3 * r + 2] * sm[2]);
    }
    return true;
}

}  // namespace svd3
}  // namespace btba
