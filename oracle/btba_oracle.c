/*
 * btba_oracle.c -- CPU restatement of BundleTrack's pose-graph bundle adjustment.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (bundletrack_amd/) never
 * links or calls anything in oracle/.
 *
 * PARITY: PINNED AGAINST THE REFERENCE ITSELF, EXECUTED ON THE CPU.  The reference (wenbowen123/BundleTrack) ships no
 * golden vectors or tests for this path and cannot be built as a GPU program here (no nvcc).  It is built as a CPU
 * program instead: oracle/ref_shim/cuda_runtime.h supplies the CUDA built-ins and a sequential kernel-launch emulator,
 * oracle/Makefile (target `ref`) streams Solver/SolverBundling.cu and the __device__ headers from /root/reference into
 * g++ (nothing is copied), and tests/test_oracle_vs_reference.py runs the reference's solveBundlingStub -- all its
 * kernels -- next to this file on identical inputs: per Gauss-Newton iterate 1e-6 on well-conditioned windows, 1.6e-5
 * at BASELINE's headline configuration c3 (bar: 1e-4); Exp / Log / computeLieUpdate / evalLie_derivI,J / bilinear taps
 * / Huber bit-identical, the 4x4 inverse value-identical, findDenseCorr accepting exactly the same pixels.  What the
 * emulation cannot reproduce is the GPU's atomic summation order (see accum_mode below).  Every function here still
 * cites the reference lines it restates (paths relative to /root/reference/src/cuda).
 *
 * Conventions
 *   - 4x4 matrices are row-major float[16] (cuda_SimpleMatrixUtil.h:1206-1215).
 *   - per-frame unknown x_k = (rot[3], trans[3]) = se(3) coordinates, T_k = Exp(x_k) maps
 *     camera k -> model; frame 0 is never moved (SolverBundling.cu:582,710,754,789).
 *   - dense JtJ is (6N)x(6N) row-major with per-frame order [trans(3), rot(3)]
 *     (SolverBundlingDenseUtil.h:251-254, SolverBundlingEquationsLie.h:115-118).
 *   - all arithmetic is IEEE fp32 like the device code, except the one double sqrt inside
 *     the Huber weight (SolverBundlingUtil.h:35) and, when accum_mode==1, the *summation*
 *     of already-rounded fp32 terms, which is carried in double and rounded once (the
 *     reference sums with float atomics in arbitrary order, so "exact sum of the fp32
 *     terms" is the order-free representative; accum_mode==0 sums sequentially in fp32).
 *
 * Build: see oracle/Makefile   (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ---- wire formats -------------------------------------------------------------------- */
/* struct EntryJ, SIFTImageManager.h:44-59: 32 bytes, invalid <=> imgIdx_i == 0xFFFFFFFF */
typedef struct {
    uint32_t imgIdx_i, imgIdx_j;
    float pos_i[3];
    float pos_j[3];
} orc_entryj;

typedef struct {
    int32_t n_gn_iters;          /* bundle.num_iter_outter = 7  (config_ycbineoat.yml:24) */
    int32_t n_pcg_iters;         /* bundle.num_iter_inner  = 5  (:25) */
    float robust_delta;          /* bundle.robust_delta 0.005   (:29) */
    float dense_dist_thresh;     /* p2p.max_dist 0.02           (:64, CUDASolverBundling.cpp:93) */
    float dense_normal_thresh;   /* cos(p2p.max_normal_angle)   (:65, CUDASolverBundling.cpp:94) */
    float depth_min, depth_max;  /* 0.1, 9999                   (CUDASolverBundling.cpp:97-98) */
    float weight_sparse;         /* 1 (SBA.cpp:28) */
    float weight_dense_depth;    /* 1 (SBA.cpp:30); 0 disables the dense term */
    int32_t accum_mode;          /* 0: sequential fp32 sums, 1: double-carried sums */
    int32_t n_threads;           /* OpenMP threads for the sweeps (<=0: library default) */
} orc_params;

/* Per-GN-iteration trace (all optional; pass NULL to skip).  Sizes for N frames, P pairs. */
typedef struct {
    float *x_after;      /* [n_gn][N][6]  (rot, trans) after the update of iteration n   */
    float *T_after;      /* [n_gn][N][16] Exp(x_after)                                  */
    float *dense_JtJ;    /* [n_gn][(6N)^2] after FlipJtJ                                 */
    float *dense_Jtr;    /* [n_gn][6N]                                                    */
    float *rhs;          /* [n_gn][N][6]  (rRot, rTrans) = PCG residual at start          */
    float *precond;      /* [n_gn][N][6]  (precRot, precTrans)                            */
    float *pcg_scalars;  /* [n_gn][n_pcg][4] = pAp, alpha, rz_new, beta                   */
    int32_t *dense_count;/* [n_gn][P] accepted dense matches per pair                     */
    float *delta;        /* [n_gn][N][6] PCG solution (deltaRot, deltaTrans)              */
} orc_trace;

/* ---- small fp32 helpers (cutil_math.h semantics) -------------------------------------- */
static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline float len3(const float a[3]) { return sqrtf(dot3(a, a)); }

/* float4x4::operator*(float4x4), cuda_SimpleMatrixUtil.h:1162-1186 */
static void m4_mul(const float *a, const float *b, float *o)
{
    float t[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            t[4 * r + c] = a[4 * r + 0] * b[0 + c] + a[4 * r + 1] * b[4 + c] + a[4 * r + 2] * b[8 + c] + a[4 * r + 3] * b[12 + c];
    memcpy(o, t, sizeof t);
}
/* float4x4::operator*(float3), w assumed 1, cuda_SimpleMatrixUtil.h:935-942 */
static inline void m4_point(const float *m, const float p[3], float o[3])
{
    float x = p[0], y = p[1], z = p[2];
    o[0] = m[0] * x + m[1] * y + m[2] * z + m[3] * 1.0f;
    o[1] = m[4] * x + m[5] * y + m[6] * z + m[7] * 1.0f;
    o[2] = m[8] * x + m[9] * y + m[10] * z + m[11] * 1.0f;
}
/* float4x4::operator*(float4), cuda_SimpleMatrixUtil.h:923-931 */
static inline void m4_vec4(const float *m, const float v[4], float o[4])
{
    float x = v[0], y = v[1], z = v[2], w = v[3];
    o[0] = m[0] * x + m[1] * y + m[2] * z + m[3] * w;
    o[1] = m[4] * x + m[5] * y + m[6] * z + m[7] * w;
    o[2] = m[8] * x + m[9] * y + m[10] * z + m[11] * w;
    o[3] = m[12] * x + m[13] * y + m[14] * z + m[15] * w;
}

/* Generic 4x4 inverse, float4x4::getInverse, cuda_SimpleMatrixUtil.h:978-1104 (the classic adjugate expansion).
 * Entry (R, C) of the adjugate is the cofactor of element (C, R): with rows r1 < r2 < r3 != C and columns
 * c1 < c2 < c3 != R, and s = (-1)^(R+C), the reference sums six triple products, each evaluated (a * b) * c, left to
 * right in exactly this order:
 *    s e[r1c1] e[r2c2] e[r3c3] - s e[r1c1] e[r2c3] e[r3c2] - s e[r2c1] e[r1c2] e[r3c3]
 *  + s e[r2c1] e[r1c3] e[r3c2] + s e[r3c1] e[r1c2] e[r2c3] - s e[r3c1] e[r1c3] e[r2c2]
 * then det = e0 adj0 + e1 adj4 + e2 adj8 + e3 adj12 (:1094) and every entry is scaled by 1/det.  Value-identical to
 * the reference's own function compiled for the CPU (tests/test_oracle_vs_reference.py; only the sign of exact zeros
 * can differ, from where the negation is applied). */
static void m4_inverse(const float *m, float *o)
{
    float adj[16];
    for (int R = 0; R < 4; R++) {
        for (int Cc = 0; Cc < 4; Cc++) {
            int r[3], c[3], a = 0, b = 0;
            for (int k = 0; k < 4; k++) { if (k != Cc) r[a++] = k; if (k != R) c[b++] = k; }
#define E(ri, ci) m[4 * r[ri] + c[ci]]
            const float t1 = E(0, 0) * E(1, 1) * E(2, 2), t2 = E(0, 0) * E(1, 2) * E(2, 1), t3 = E(1, 0) * E(0, 1) * E(2, 2);
            const float t4 = E(1, 0) * E(0, 2) * E(2, 1), t5 = E(2, 0) * E(0, 1) * E(1, 2), t6 = E(2, 0) * E(0, 2) * E(1, 1);
#undef E
            const float v = ((R + Cc) & 1) ? (((((-t1) + t2) + t3) - t4) - t5) + t6 : ((((t1 - t2) - t3) + t4) + t5) - t6;
            adj[4 * R + Cc] = v;
        }
    }
    float det = m[0] * adj[0] + m[1] * adj[4] + m[2] * adj[8] + m[3] * adj[12];
    float rdet = 1.0f / det;
    for (int k = 0; k < 16; k++) o[k] = adj[k] * rdet;
}

/* ---- SE(3) / SO(3), LieDerivUtil.h ----------------------------------------------------- */
#define ONE_TWENTIETH 0.05f
#define ONE_SIXTH 0.16666667f

/* rodrigues_so3_exp, LieDerivUtil.h:17-45; R row-major 3x3 */
static void rodrigues(const float w[3], float A, float B, float R[9])
{
    float wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
    R[0] = 1.0f - B * (wy2 + wz2);
    R[4] = 1.0f - B * (wx2 + wz2);
    R[8] = 1.0f - B * (wx2 + wy2);
    float a = A * w[2], b = B * (w[0] * w[1]);
    R[1] = b - a; R[3] = b + a;
    a = A * w[1]; b = B * (w[0] * w[2]);
    R[2] = b + a; R[6] = b - a;
    a = A * w[0]; b = B * (w[1] * w[2]);
    R[5] = b - a; R[7] = b + a;
}
/* exp_rotation, LieDerivUtil.h:47-71 */
static void exp_rotation(const float w[3], float R[9])
{
    float theta_sq = dot3(w, w);
    float theta = sqrtf(theta_sq);
    float A, B;
    if ((double)theta_sq < 1e-8) {
        A = 1.0f - ONE_SIXTH * theta_sq;
        B = 0.5f;
    } else if ((double)theta_sq < 1e-6) {
        B = 0.5f - 0.25f * ONE_SIXTH * theta_sq;
        A = 1.0f - theta_sq * ONE_SIXTH * (1.0f - ONE_TWENTIETH * theta_sq);
    } else {
        float inv_theta = 1.0f / theta;
        A = sinf(theta) * inv_theta;
        B = (1 - cosf(theta)) * (inv_theta * inv_theta);
    }
    rodrigues(w, A, B, R);
}
/* ln_rotation, LieDerivUtil.h:73-124 */
static void ln_rotation(const float R[9], float out[3])
{
    float cos_angle = ((R[0] + R[4] + R[8]) - 1.0f) * 0.5f;
    float r[3] = { (R[7] - R[5]) * 0.5f, (R[2] - R[6]) * 0.5f, (R[3] - R[1]) * 0.5f };
    float sin_angle_abs = len3(r);
    if (cos_angle > (float)0.70710678118654752440) {
        if (sin_angle_abs > 0) {
            float s = asinf(sin_angle_abs) / sin_angle_abs;
            r[0] *= s; r[1] *= s; r[2] *= s;
        }
    } else if (cos_angle > -(float)0.70710678118654752440) {
        float angle = acosf(cos_angle);
        float s = angle / sin_angle_abs;
        r[0] *= s; r[1] *= s; r[2] *= s;
    } else {
        const float angle = 3.14159265358979323846f - asinf(sin_angle_abs);
        const float d0 = R[0] - cos_angle, d1 = R[4] - cos_angle, d2 = R[8] - cos_angle;
        float r2[3];
        if (fabsf(d0) > fabsf(d1) && fabsf(d0) > fabsf(d2)) {
            r2[0] = d0; r2[1] = (R[3] + R[1]) * 0.5f; r2[2] = (R[2] + R[6]) * 0.5f;
        } else if (fabsf(d1) > fabsf(d2)) {
            r2[0] = (R[3] + R[1]) * 0.5f; r2[1] = d1; r2[2] = (R[7] + R[5]) * 0.5f;
        } else {
            r2[0] = (R[2] + R[6]) * 0.5f; r2[1] = (R[7] + R[5]) * 0.5f; r2[2] = d2;
        }
        if (dot3(r2, r) < 0) { r2[0] *= -1; r2[1] *= -1; r2[2] *= -1; }
        float s = angle / len3(r2);
        r[0] = r2[0] * s; r[1] = r2[1] * s; r[2] = r2[2] * s;
    }
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}
/* matrixToPose, LieDerivUtil.h:126-148 */
ORC_API void orc_matrix_to_pose(const float *M, float rot[3], float trans[3])
{
    float R[9] = { M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10] };
    float t[3] = { M[3], M[7], M[11] };
    ln_rotation(R, rot);
    const float theta = len3(rot);
    float shtot = 0.5f;
    if (theta > 0.00001f) shtot = sinf(theta * 0.5f) / theta;
    float rot_half[3] = { rot[0] * -0.5f, rot[1] * -0.5f, rot[2] * -0.5f };
    float H[9];
    exp_rotation(rot_half, H);
    float tr[3] = { H[0] * t[0] + H[1] * t[1] + H[2] * t[2],
                    H[3] * t[0] + H[4] * t[1] + H[5] * t[2],
                    H[6] * t[0] + H[7] * t[1] + H[8] * t[2] };
    float s;
    if (theta > 0.001f) s = dot3(t, rot) * (1 - 2 * shtot) / dot3(rot, rot);
    else s = dot3(t, rot) / 24;
    tr[0] -= rot[0] * s; tr[1] -= rot[1] * s; tr[2] -= rot[2] * s;
    float k = 1.0f / (2 * shtot);
    trans[0] = tr[0] * k; trans[1] = tr[1] * k; trans[2] = tr[2] * k;
}
/* poseToMatrix, LieDerivUtil.h:150-194 */
ORC_API void orc_pose_to_matrix(const float rot[3], const float trans[3], float *M)
{
    float translation[3], R[9];
    const float theta_sq = dot3(rot, rot);
    const float theta = sqrtf(theta_sq);
    float A, B;
    float cr[3];
    cross3(rot, trans, cr);
    if ((double)theta_sq < 1e-8) {
        A = 1.0f - ONE_SIXTH * theta_sq;
        B = 0.5f;
        for (int k = 0; k < 3; k++) translation[k] = trans[k] + 0.5f * cr[k];
    } else {
        float C;
        if ((double)theta_sq < 1e-6) {
            C = ONE_SIXTH * (1.0f - ONE_TWENTIETH * theta_sq);
            A = 1.0f - theta_sq * C;
            B = 0.5f - 0.25f * ONE_SIXTH * theta_sq;
        } else {
            const float inv_theta = 1.0f / theta;
            A = sinf(theta) * inv_theta;
            B = (1 - cosf(theta)) * (inv_theta * inv_theta);
            C = (1 - A) * (inv_theta * inv_theta);
        }
        float wc[3];
        cross3(rot, cr, wc);
        for (int k = 0; k < 3; k++) translation[k] = trans[k] + B * cr[k] + C * wc[k];
    }
    rodrigues(rot, A, B, R);
    M[0] = R[0]; M[1] = R[1]; M[2] = R[2];  M[3] = translation[0];
    M[4] = R[3]; M[5] = R[4]; M[6] = R[5];  M[7] = translation[1];
    M[8] = R[6]; M[9] = R[7]; M[10] = R[8]; M[11] = translation[2];
    M[12] = 0.0f; M[13] = 0.0f; M[14] = 0.0f; M[15] = 1.0f;
}
ORC_API void orc_mat4_inverse(const float *m, float *o) { m4_inverse(m, o); }

/* computeLieUpdate, LieDerivUtil.h:276-282 */
static void lie_update(const float dW[3], const float dT[3], const float cW[3], const float cT[3], float nW[3], float nT[3])
{
    float U[16], C[16], P[16];
    orc_pose_to_matrix(dW, dT, U);
    orc_pose_to_matrix(cW, cT, C);
    m4_mul(U, C, P);
    orc_matrix_to_pose(P, nW, nT);
}

/* huberLoss -> rho.y, SolverBundlingUtil.h:24-40 (float sqrt stored in a double, division in double) */
static inline float huber_weight(float e, float delta)
{
    float dsqr = delta * delta;
    if (e <= dsqr) return 1.0f;
    double sqrte = sqrtf(e);
    return (float)(delta / sqrte);
}

/* ---- frame cache, CUDACache.cpp:14-38,76-88 + CUDAImageUtil.cu:50-99,310-327,965-976 ---- */
/* K row-major 3x3 full-resolution intrinsics.  Outputs at Wd x Hd:
 *   campos[npix][4], normals_d[npix][4], depth_d[npix], *n_valid, intr_d = (fx,fy,cx,cy) scaled. */
ORC_API int orc_build_cache(int H, int W, int Hd, int Wd, const float *K,
                            const float *depth, const float *normals,
                            float *campos, float *normals_d, float *depth_d,
                            int32_t *n_valid, float *intr_d)
{
    if (H <= 1 || W <= 1 || Hd <= 1 || Wd <= 1) return -1;
    /* CUDACache.cpp:20-24 */
    if (intr_d) {
        intr_d[0] = K[0] * ((float)Wd / (float)W);
        intr_d[1] = K[4] * ((float)Hd / (float)H);
        intr_d[2] = K[2] * ((float)(Wd - 1) / (float)(W - 1));
        intr_d[3] = K[5] * ((float)(Hd - 1) / (float)(H - 1));
    }
    /* m_inputIntrinsicsInv = inverse of [fx 0 cx 0; 0 fy cy 0; 0 0 1 0; 0 0 0 1] (CUDACache.cpp:33).
     * Computed with the same generic cofactor inverse the device math uses. */
    float Kin[16] = { K[0], K[1], K[2], 0, K[3], K[4], K[5], 0, K[6], K[7], K[8], 0, 0, 0, 0, 1 };
    float Ki[16];
    m4_inverse(Kin, Ki);
    /* resample*_Kernel: scale and rounding, CUDAImageUtil.cu:57-61 */
    const float scaleW = (float)(W - 1) / (float)(Wd - 1);
    const float scaleH = (float)(H - 1) / (float)(Hd - 1);
    int cnt = 0;
    for (int y = 0; y < Hd; y++) {
        for (int x = 0; x < Wd; x++) {
            unsigned xi = (unsigned)(x * scaleW + 0.5f);
            unsigned yi = (unsigned)(y * scaleH + 0.5f);
            int o = y * Wd + x;
            if (!(xi < (unsigned)W && yi < (unsigned)H)) continue; /* never for Hd<=H */
            size_t s = (size_t)yi * W + xi;
            float d = depth[s];
            /* convertDepthFloatToCameraSpaceFloat4_Kernel, CUDAImageUtil.cu:310-327 */
            float cp[4] = { 0, 0, 0, 0 };
            if (d >= 0.1) {
                float v[4] = { (float)xi * d, (float)yi * d, d, d }, c[4];
                m4_vec4(Ki, v, c);
                cp[0] = c[0]; cp[1] = c[1]; cp[2] = c[3]; cp[3] = 1.0f;
            }
            memcpy(campos + 4 * o, cp, sizeof cp);
            memcpy(normals_d + 4 * o, normals + 4 * s, 4 * sizeof(float));
            if (depth_d) depth_d[o] = d;
            if (d >= 0.1) cnt++;                      /* countNumValidDepth_Kernel :965-976 */
        }
    }
    if (n_valid) *n_valid = cnt;
    return 0;
}

/* ---- dense term ------------------------------------------------------------------------ */
/* bilinearInterpolationFloat4, ICPUtil.h:83-110.  MINF sentinel = -inf. */
static int bilinear4(float x, float y, const float *img, unsigned W, unsigned H, float out[4])
{
    const int p00x = (int)floorf(x), p00y = (int)floorf(y);
    const int p01x = p00x, p01y = p00y + 1;
    const int p10x = p00x + 1, p10y = p00y;
    const int p11x = p00x + 1, p11y = p00y + 1;
    const float alpha = x - p00x;
    const float beta = y - p00y;
    const float MINF = -INFINITY;
    float s0[4] = { 0, 0, 0, 0 }, w0 = 0.0f;
    if ((unsigned)p00x < W && (unsigned)p00y < H) { const float *v = img + 4 * ((size_t)p00y * W + p00x); if (v[0] != MINF) { for (int k = 0; k < 4; k++) s0[k] += (1.0f - alpha) * v[k]; w0 += (1.0f - alpha); } }
    if ((unsigned)p10x < W && (unsigned)p10y < H) { const float *v = img + 4 * ((size_t)p10y * W + p10x); if (v[0] != MINF) { for (int k = 0; k < 4; k++) s0[k] += alpha * v[k]; w0 += alpha; } }
    float s1[4] = { 0, 0, 0, 0 }, w1 = 0.0f;
    if ((unsigned)p01x < W && (unsigned)p01y < H) { const float *v = img + 4 * ((size_t)p01y * W + p01x); if (v[0] != MINF) { for (int k = 0; k < 4; k++) s1[k] += (1.0f - alpha) * v[k]; w1 += (1.0f - alpha); } }
    if ((unsigned)p11x < W && (unsigned)p11y < H) { const float *v = img + 4 * ((size_t)p11y * W + p11x); if (v[0] != MINF) { for (int k = 0; k < 4; k++) s1[k] += alpha * v[k]; w1 += alpha; } }
    float ss[4] = { 0, 0, 0, 0 }, ww = 0.0f;
    if (w0 > 0.0f) { for (int k = 0; k < 4; k++) ss[k] += (1.0f - beta) * (s0[k] / w0); ww += (1.0f - beta); }
    if (w1 > 0.0f) { for (int k = 0; k < 4; k++) ss[k] += beta * (s1[k] / w1); ww += beta; }
    if (ww > 0.0f) { for (int k = 0; k < 4; k++) out[k] = ss[k] / ww; return 1; }
    out[0] = out[1] = out[2] = out[3] = MINF;
    return 0;
}

/* findDenseCorr (float4 camPos / float4 normal variant), SolverBundlingDenseUtil.h:78-110 */
static int find_dense_corr(unsigned idx, unsigned W, unsigned H, float distThresh, float normalThresh,
                           const float *T /*tgt<-src*/, const float intr[4],
                           const float *tgtCam, const float *tgtNrm, const float *srcCam, const float *srcNrm,
                           float depthMin, float depthMax,
                           float camPosSrc[3], float camPosSrcToTgt[3], float camPosTgt[3], float normalTgt[3])
{
    const float *cposj = srcCam + 4 * (size_t)idx;
    if (!(cposj[2] > depthMin && cposj[2] < depthMax)) return 0;
    camPosSrc[0] = cposj[0]; camPosSrc[1] = cposj[1]; camPosSrc[2] = cposj[2];
    float nrmj[4];
    memcpy(nrmj, srcNrm + 4 * (size_t)idx, sizeof nrmj);
    if (!(nrmj[0] != -INFINITY)) return 0;
    float nt[4];
    m4_vec4(T, nrmj, nt);
    m4_point(T, camPosSrc, camPosSrcToTgt);
    /* cameraToDepth, CUDACameraUtil.h:9-14 */
    float u = camPosSrcToTgt[0] * intr[0] / camPosSrcToTgt[2] + intr[2];
    float v = camPosSrcToTgt[1] * intr[1] / camPosSrcToTgt[2] + intr[3];
    int sx = (int)roundf(u), sy = (int)roundf(v);
    if (!(sx >= 0 && sy >= 0 && sx < (int)W && sy < (int)H)) return 0;
    float cposi[4];
    bilinear4(u, v, tgtCam, W, H, cposi);
    if (!(cposi[2] > depthMin && cposi[2] < depthMax)) return 0;
    camPosTgt[0] = cposi[0]; camPosTgt[1] = cposi[1]; camPosTgt[2] = cposi[2];
    float nrmi[4];
    bilinear4(u, v, tgtNrm, W, H, nrmi);
    if (!(nrmi[0] != -INFINITY)) return 0;
    normalTgt[0] = nrmi[0]; normalTgt[1] = nrmi[1]; normalTgt[2] = nrmi[2];
    float d[3] = { camPosSrcToTgt[0] - camPosTgt[0], camPosSrcToTgt[1] - camPosTgt[1], camPosSrcToTgt[2] - camPosTgt[2] };
    float dist = len3(d);
    float dNormal = nt[0] * nrmi[0] + nt[1] * nrmi[1] + nt[2] * nrmi[2] + nt[3] * nrmi[3];
    return (dNormal >= normalThresh && dist <= distThresh);
}

/* matNxM::operator*, cuda_SimpleMatrixUtil.h:1349-1372 (sum from 0, k ascending) */
static void mat_mul(const float *a, int n, int m, const float *b, int mo, float *o)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < mo; j++) {
            float sum = 0.0f;
            for (int k = 0; k < m; k++) sum += a[i * m + k] * b[k * mo + j];
            o[i * mo + j] = sum;
        }
}
/* evalLie_derivI, LieDerivUtil.h:228-253 -- literal 3x12 * 12x6 product */
static void lie_deriv_I(const float *A, const float *D, const float p[3], float jac[18])
{
    float T[16];
    m4_mul(A, D, T);
    float pt[3] = { p[0] - T[3], p[1] - T[7], p[2] - T[11] };
    float j0[36], j1[72];
    memset(j0, 0, sizeof j0); memset(j1, 0, sizeof j1);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) j0[r * 12 + 3 * r + c] = pt[c];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            j0[r * 12 + c + 9] = -T[4 * c + r];
            j1[(r + 9) * 6 + c] = A[4 * r + c];
        }
    float RA[9] = { A[0], A[1], A[2], A[4], A[5], A[6], A[8], A[9], A[10] };
    for (int k = 0; k < 4; k++) {
        float v[3] = { D[k], D[4 + k], D[8 + k] };
        /* VectorToSkewSymmetricMatrix, LieDerivUtil.h:203-212 */
        float S[9] = { 0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0 };
        float m[9];
        mat_mul(RA, 3, 3, S, 3, m);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) j1[(3 * k + r) * 6 + 3 + c] = m[3 * r + c] * -1.0f;
    }
    mat_mul(j0, 3, 12, j1, 6, jac);
}
/* evalLie_derivJ, LieDerivUtil.h:255-273 */
static void lie_deriv_J(const float *A, const float *D, const float p[3], float jac[18])
{
    float dr1[3] = { D[0], D[1], D[2] }, dr2[3] = { D[4], D[5], D[6] }, dr3[3] = { D[8], D[9], D[10] };
    float dtx = D[3], dty = D[7], dtz = D[11];
    float j[18];
    j[0] = 1.0f; j[1] = 0.0f; j[2] = 0.0f;
    j[6] = 0.0f; j[7] = 1.0f; j[8] = 0.0f;
    j[12] = 0.0f; j[13] = 0.0f; j[14] = 1.0f;
    j[3] = 0.0f;                       j[4] = dot3(p, dr3) + dtz;      j[5] = -(dot3(p, dr2) + dty);
    j[9] = -(dot3(p, dr3) + dtz);      j[10] = 0.0f;                   j[11] = dot3(p, dr1) + dtx;
    j[15] = dot3(p, dr2) + dty;        j[16] = -(dot3(p, dr1) + dtx);  j[17] = 0.0f;
    float RA[9] = { A[0], A[1], A[2], A[4], A[5], A[6], A[8], A[9], A[10] };
    mat_mul(RA, 3, 3, j, 6, jac);
}

/* Summation helper: fp32 terms summed either in float (sequential) or carried in double. */
typedef struct { double *d; float *f; int n; int mode; } accum_t;
static void acc_init(accum_t *a, int n, int mode) { a->n = n; a->mode = mode; a->d = (double *)calloc(n, sizeof(double)); a->f = (float *)calloc(n, sizeof(float)); }
static inline void acc_add(accum_t *a, int k, float term) { if (a->mode) a->d[k] += (double)term; else a->f[k] += term; }
static inline float acc_get(const accum_t *a, int k) { return a->mode ? (float)a->d[k] : a->f[k]; }
static void acc_free(accum_t *a) { free(a->d); free(a->f); }
static void acc_merge(accum_t *dst, const accum_t *src) { for (int k = 0; k < dst->n; k++) { dst->d[k] += src->d[k]; dst->f[k] += src->f[k]; } }

/* BuildDenseSystem_Kernel<true,false> + addToLocalSystem + FlipJtJ_Kernel,
 * SolverBundling.cu:129-229, 49-59; SolverBundlingDenseUtil.h:217-275.
 * pairs[2*p] = i (target), pairs[2*p+1] = j (source): the ordered dense pair list the
 * reference derives from allocator addresses (SolverBundling.cu:17-47) is an
 * explicit input here. */
static void build_dense_system(const orc_params *prm, int N, int Wd, int Hd, const float intr[4],
                               const float *campos, const float *normals,
                               const int32_t *pairs, int P,
                               const float *T, const float *Tinv,
                               float *JtJ, float *Jtr, int32_t *counts)
{
    const int dim = 6 * N, npix = Wd * Hd;
    const size_t fstride = (size_t)npix * 4;
    int nthr = 1;
#ifdef _OPENMP
    nthr = prm->n_threads > 0 ? prm->n_threads : omp_get_max_threads();
#endif
    accum_t *accs = (accum_t *)malloc(sizeof(accum_t) * nthr);
    for (int t = 0; t < nthr; t++) acc_init(&accs[t], dim * dim + dim, prm->accum_mode);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthr)
    for (int p = 0; p < P; p++) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        accum_t *acc = &accs[tid];
        const unsigned i = (unsigned)pairs[2 * p], j = (unsigned)pairs[2 * p + 1];
        const float *Ti = T + 16 * i, *Tj = T + 16 * j, *Tii = Tinv + 16 * i, *Tji = Tinv + 16 * j;
        float Tij[16];
        m4_mul(Tii, Tj, Tij);                                   /* :148 */
        int cnt = 0;
        for (int s = 0; s < npix; s++) {
            float cs[3], cst[3], ct[3], nt[3];
            if (!find_dense_corr((unsigned)s, (unsigned)Wd, (unsigned)Hd, prm->dense_dist_thresh, prm->dense_normal_thresh,
                                 Tij, intr, campos + fstride * i, normals + fstride * i,
                                 campos + fstride * j, normals + fstride * j,
                                 prm->depth_min, prm->depth_max, cs, cst, ct, nt))
                continue;
            cnt++;
            float diff[3] = { ct[0] - cst[0], ct[1] - cst[1], ct[2] - cst[2] };
            float res = dot3(diff, nt);                          /* :180-181 */
            float w = prm->weight_dense_depth * huber_weight(res * res, prm->robust_delta); /* :183-184 */
            float ri[6] = { 0, 0, 0, 0, 0, 0 }, rj[6] = { 0, 0, 0, 0, 0, 0 }, jac[18];
            if (i > 0) {                                         /* computeJacobianBlockRow_i, Lie.h:214-221 */
                lie_deriv_I(Tji, Ti, cs, jac);
                for (int c = 0; c < 6; c++) { float col[3] = { jac[c], jac[6 + c], jac[12 + c] }; ri[c] = -dot3(col, nt); }
            }
            if (j > 0) {                                         /* computeJacobianBlockRow_j, Lie.h:223-230 */
                lie_deriv_J(Tii, Tj, cs, jac);
                for (int c = 0; c < 6; c++) { float col[3] = { jac[c], jac[6 + c], jac[12 + c] }; rj[c] = -dot3(col, nt); }
            }
            /* addToLocalSystem, SolverBundlingDenseUtil.h:221-275 */
            for (int a = 0; a < 6; a++) {
                for (int b = a; b < 6; b++) {
                    if (i > 0) acc_add(acc, (i * 6 + b) * dim + (i * 6 + a), ri[a] * ri[b] * w);
                    if (j > 0) acc_add(acc, (j * 6 + b) * dim + (j * 6 + a), rj[a] * rj[b] * w);
                    if (i > 0 && j > 0) {
                        acc_add(acc, (j * 6 + b) * dim + (i * 6 + a), ri[a] * rj[b] * w);
                        if (a != b) acc_add(acc, (j * 6 + a) * dim + (i * 6 + b), ri[b] * rj[a] * w);
                    }
                }
                if (i > 0) acc_add(acc, dim * dim + i * 6 + a, ri[a] * res * w);
                if (j > 0) acc_add(acc, dim * dim + j * 6 + a, rj[a] * res * w);
            }
        }
        if (counts) counts[p] = cnt;
    }
    for (int t = 1; t < nthr; t++) acc_merge(&accs[0], &accs[t]);
    for (int k = 0; k < dim * dim; k++) JtJ[k] = acc_get(&accs[0], k);
    for (int k = 0; k < dim; k++) Jtr[k] = acc_get(&accs[0], dim * dim + k);
    for (int t = 0; t < nthr; t++) acc_free(&accs[t]);
    free(accs);
    /* FlipJtJ_Kernel, SolverBundling.cu:49-59: upper <- lower */
    for (int y = 0; y < dim; y++)
        for (int x = y + 1; x < dim; x++) JtJ[y * dim + x] = JtJ[x * dim + y];
}

/* ---- sparse term ----------------------------------------------------------------------- */
/* Frame -> correspondence table, BuildVariablesToCorrespondencesTableDevice,
 * SolverBundling.cu:1006-1031.  Entries are appended in correspondence order (the device
 * order is nondeterministic); overflow beyond max_per_image invalidates the correspondence. */
typedef struct { int *rows; int *counts; int stride; } corr_table;

static void build_table(orc_entryj *corr, int C, int N, int max_per_image, corr_table *t)
{
    t->stride = max_per_image;
    t->rows = (int *)malloc(sizeof(int) * (size_t)N * (max_per_image > 0 ? max_per_image : 1));
    t->counts = (int *)calloc(N, sizeof(int));
    for (int x = 0; x < C; x++) {
        orc_entryj *c = &corr[x];
        if (c->imgIdx_i == 0xFFFFFFFFu) continue;
        int o0 = t->counts[c->imgIdx_i]++;
        int o1 = t->counts[c->imgIdx_j]++;
        if (o0 < max_per_image && o1 < max_per_image) {
            t->rows[(size_t)c->imgIdx_i * max_per_image + o0] = x;
            t->rows[(size_t)c->imgIdx_j * max_per_image + o1] = x;
        } else {
            c->imgIdx_i = 0xFFFFFFFFu; c->imgIdx_j = 0xFFFFFFFFu;
        }
    }
}

/* evalMinusJTFDevice<useDense>, SolverBundlingEquationsLie.h:60-137 */
static void eval_minus_JTF(const orc_params *prm, int k, const orc_entryj *corr, const corr_table *tab, const float *T,
                           int use_dense, const float *Jtr, float resRot[3], float resTrans[3], float precRot[3], float precTrans[3])
{
    accum_t acc;
    acc_init(&acc, 12, prm->accum_mode);
    int n = tab->counts[k] < tab->stride ? tab->counts[k] : tab->stride;
    for (int q = 0; q < n; q++) {
        const orc_entryj *c = &corr[tab->rows[(size_t)k * tab->stride + q]];
        if (c->imgIdx_i == 0xFFFFFFFFu) continue;
        const float *TI = T + 16 * c->imgIdx_i, *TJ = T + 16 * c->imgIdx_j;
        float wp[3], sign = 1;
        if ((unsigned)k != c->imgIdx_i) { sign = -1; m4_point(TJ, c->pos_j, wp); }
        else m4_point(TI, c->pos_i, wp);
        /* evalLie_dAlpha/dBeta/dGamma, LieDerivUtil.h:215-226 */
        float da[3] = { 0.0f, -wp[2], wp[1] }, db[3] = { wp[2], 0.0f, -wp[0] }, dc[3] = { -wp[1], wp[0], 0.0f };
        float a[3], b[3];
        m4_point(TI, c->pos_i, a); m4_point(TJ, c->pos_j, b);
        float r[3] = { a[0] - b[0], a[1] - b[1], a[2] - b[2] };
        float rho = huber_weight(dot3(r, r), prm->robust_delta);
        acc_add(&acc, 0, rho * sign * dot3(da, r)); acc_add(&acc, 1, rho * sign * dot3(db, r)); acc_add(&acc, 2, rho * sign * dot3(dc, r));
        acc_add(&acc, 3, rho * sign * r[0]); acc_add(&acc, 4, rho * sign * r[1]); acc_add(&acc, 5, rho * sign * r[2]);
        acc_add(&acc, 6, rho * dot3(da, da)); acc_add(&acc, 7, rho * dot3(db, db)); acc_add(&acc, 8, rho * dot3(dc, dc));
        acc_add(&acc, 9, rho * 1.0f); acc_add(&acc, 10, rho * 1.0f); acc_add(&acc, 11, rho * 1.0f);
    }
    for (int q = 0; q < 3; q++) {
        resRot[q] = -prm->weight_sparse * acc_get(&acc, q);
        resTrans[q] = -prm->weight_sparse * acc_get(&acc, 3 + q);
    }
    if (use_dense) {
        for (int q = 0; q < 3; q++) { resRot[q] -= Jtr[k * 6 + 3 + q]; resTrans[q] -= Jtr[k * 6 + q]; }
    }
    for (int q = 0; q < 3; q++) {
        float pr = acc_get(&acc, 6 + q), pt = acc_get(&acc, 9 + q);
        precRot[q] = (pr > 0.000001f) ? 1.0f / pr : 1.0f;
        precTrans[q] = (pt > 0.000001f) ? 1.0f / pt : 1.0f;
    }
    acc_free(&acc);
}

/* applyJDevice, SolverBundlingEquationsLie.h:179-211 */
static void apply_J(const orc_params *prm, const orc_entryj *c, const float *T, const float *pRot, const float *pTrans, float out[3])
{
    float b[3] = { 0, 0, 0 };
    if (c->imgIdx_i != 0xFFFFFFFFu) {
        if (c->imgIdx_i > 0) {
            float wp[3];
            m4_point(T + 16 * c->imgIdx_i, c->pos_i, wp);
            float da[3] = { 0.0f, -wp[2], wp[1] }, db[3] = { wp[2], 0.0f, -wp[0] }, dc[3] = { -wp[1], wp[0], 0.0f };
            const float *pp = pRot + 3 * c->imgIdx_i, *pt = pTrans + 3 * c->imgIdx_i;
            for (int q = 0; q < 3; q++) b[q] += da[q] * pp[0] + db[q] * pp[1] + dc[q] * pp[2] + pt[q];
        }
        if (c->imgIdx_j > 0) {
            float wp[3];
            m4_point(T + 16 * c->imgIdx_j, c->pos_j, wp);
            float da[3] = { 0.0f, -wp[2], wp[1] }, db[3] = { wp[2], 0.0f, -wp[0] }, dc[3] = { -wp[1], wp[0], 0.0f };
            const float *pp = pRot + 3 * c->imgIdx_j, *pt = pTrans + 3 * c->imgIdx_j;
            for (int q = 0; q < 3; q++) b[q] -= da[q] * pp[0] + db[q] * pp[1] + dc[q] * pp[2] + pt[q];
        }
        for (int q = 0; q < 3; q++) b[q] *= prm->weight_sparse;
    }
    out[0] = b[0]; out[1] = b[1]; out[2] = b[2];
}

/* applyJTDevice, SolverBundlingEquationsLie.h:140-177 (per frame, over its table row) */
static void apply_JT(const orc_params *prm, int k, const orc_entryj *corr, const corr_table *tab, const float *T, const float *Jp,
                     float outRot[3], float outTrans[3])
{
    accum_t acc;
    acc_init(&acc, 6, prm->accum_mode);
    int n = tab->counts[k] < tab->stride ? tab->counts[k] : tab->stride;
    for (int q = 0; q < n; q++) {
        int ci = tab->rows[(size_t)k * tab->stride + q];
        const orc_entryj *c = &corr[ci];
        if (c->imgIdx_i == 0xFFFFFFFFu) continue;
        float wp[3], sign = 1;
        if ((unsigned)k != c->imgIdx_i) { sign = -1; m4_point(T + 16 * c->imgIdx_j, c->pos_j, wp); }
        else m4_point(T + 16 * c->imgIdx_i, c->pos_i, wp);
        float da[3] = { 0.0f, -wp[2], wp[1] }, db[3] = { wp[2], 0.0f, -wp[0] }, dc[3] = { -wp[1], wp[0], 0.0f };
        const float *jp = Jp + 3 * (size_t)ci;
        acc_add(&acc, 0, sign * dot3(da, jp)); acc_add(&acc, 1, sign * dot3(db, jp)); acc_add(&acc, 2, sign * dot3(dc, jp));
        acc_add(&acc, 3, sign * jp[0]); acc_add(&acc, 4, sign * jp[1]); acc_add(&acc, 5, sign * jp[2]);
    }
    for (int q = 0; q < 3; q++) { outRot[q] = acc_get(&acc, q); outTrans[q] = acc_get(&acc, 3 + q); }
    acc_free(&acc);
}

/* applyJTJDenseDevice, SolverBundlingDenseUtil.h:349-385: sum over frames m = 1..N-1 */
static void apply_JTJ_dense(int k, int N, const float *JtJ, const float *pRot, const float *pTrans, float outRot[3], float outTrans[3])
{
    const int dim = 6 * N, bv = 6 * k;
    outRot[0] = outRot[1] = outRot[2] = 0.0f; outTrans[0] = outTrans[1] = outTrans[2] = 0.0f;
    for (int m = 1; m < N; m++) {
        const int bi = 6 * m;
        const float *pt = pTrans + 3 * m, *pr = pRot + 3 * m;
        for (int r = 0; r < 3; r++) {
            const float *row = JtJ + (size_t)(bv + r) * dim + bi;
            const float *row3 = JtJ + (size_t)(bv + 3 + r) * dim + bi;
            float b00 = row[0] * pt[0] + row[1] * pt[1] + row[2] * pt[2];
            float b01 = row[3] * pr[0] + row[4] * pr[1] + row[5] * pr[2];
            float b10 = row3[0] * pt[0] + row3[1] * pt[1] + row3[2] * pt[2];
            float b11 = row3[3] * pr[0] + row3[4] * pr[1] + row3[5] * pr[2];
            outTrans[r] += b00 + b01;
            outRot[r] += b10 + b11;
        }
    }
}

/* ---- the solver: solveBundlingStub, SolverBundling.cu:931-1003 ---------------------------
 * campos / normals: [N][npix][4] at the downsampled resolution (CUDACachedFrame buffers),
 * intr = (fx, fy, cx, cy) downscaled, corr: C EntryJ (copied; overflow may invalidate),
 * pairs: P ordered (target, source) dense pairs, poses: [N][16] row-major in/out
 * (SBA::align, SBA.cpp:106-115: Log before, Exp after).  Returns 0 on success. */
/* Many independent instances at once, one single-threaded solve per OpenMP thread (bench.py's all-core CPU baseline: the
 * c5 configuration shards instances, so the CPU gets the same parallelisation).  Job j solves instance j % B from a private copy of
 * its poses; results are discarded, the return value is the number of solves completed. */
ORC_API int orc_solve(const orc_params *prm, int N, int Wd, int Hd, const float *intr, const float *campos, const float *normals,
                      const orc_entryj *corr_in, int C, const int32_t *pairs, int P, float *poses, orc_trace *tr);
ORC_API int orc_solve_jobs(const orc_params *prm_in, int n_jobs, int n_workers, int B, int N, int Wd, int Hd, const float *intr,
                           const float *const *campos, const float *const *normals, const orc_entryj *const *corr, const int32_t *C,
                           const int32_t *pairs, int P, const float *const *poses)
{
    int done = 0;
    orc_params prm = *prm_in;
    prm.n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_workers > 0 ? n_workers : 1) reduction(+ : done)
    for (int j = 0; j < n_jobs; j++) {
        const int b = j % B;
        float *p = (float *)malloc(sizeof(float) * 16 * (size_t)N);
        memcpy(p, poses[b], sizeof(float) * 16 * (size_t)N);
        if (orc_solve(&prm, N, Wd, Hd, intr + 4 * b, campos[b], normals[b], corr[b], C[b], pairs, P, p, NULL) == 0) done++;
        free(p);
    }
    return done;
}

ORC_API int orc_solve(const orc_params *prm, int N, int Wd, int Hd, const float *intr,
                      const float *campos, const float *normals,
                      const orc_entryj *corr_in, int C,
                      const int32_t *pairs, int P,
                      float *poses, orc_trace *tr)
{
    if (N < 2 || prm->n_gn_iters < 1) return -1;              /* MLIB_ASSERT, CUDASolverBundling.cpp:194 */
    const int dim = 6 * N;
    orc_entryj *corr = (orc_entryj *)malloc(sizeof(orc_entryj) * (C > 0 ? C : 1));
    if (C > 0) memcpy(corr, corr_in, sizeof(orc_entryj) * C);
    /* max_corr_per_image, LossGPU.cu:104-115 */
    int *per = (int *)calloc(N, sizeof(int));
    for (int x = 0; x < C; x++) { if (corr[x].imgIdx_i < (unsigned)N) per[corr[x].imgIdx_i]++; if (corr[x].imgIdx_j < (unsigned)N) per[corr[x].imgIdx_j]++; }
    int max_per = 0;
    for (int k = 0; k < N; k++) if (per[k] > max_per) max_per = per[k];
    free(per);
    corr_table tab;
    build_table(corr, C, N, max_per, &tab);                    /* CUDASolverBundling.cpp:259-262 */

    float *xRot = (float *)calloc(3 * N, sizeof(float)), *xTrans = (float *)calloc(3 * N, sizeof(float));
    float *T = (float *)malloc(sizeof(float) * 16 * N), *Tinv = (float *)malloc(sizeof(float) * 16 * N);
    float *JtJ = (float *)calloc((size_t)dim * dim, sizeof(float)), *Jtr = (float *)calloc(dim, sizeof(float));
    float *rRot = (float *)calloc(3 * N, 4), *rTrans = (float *)calloc(3 * N, 4), *zRot = (float *)calloc(3 * N, 4), *zTrans = (float *)calloc(3 * N, 4);
    float *pRot = (float *)calloc(3 * N, 4), *pTrans = (float *)calloc(3 * N, 4), *dRot = (float *)calloc(3 * N, 4), *dTrans = (float *)calloc(3 * N, 4);
    float *ApRot = (float *)calloc(3 * N, 4), *ApTrans = (float *)calloc(3 * N, 4), *mRot = (float *)calloc(3 * N, 4), *mTrans = (float *)calloc(3 * N, 4);
    float *Jp = (float *)calloc(3 * (size_t)(C > 0 ? C : 1), sizeof(float));
    int32_t *cnts = (int32_t *)calloc(P > 0 ? P : 1, sizeof(int32_t));
    int nthr = 1;
#ifdef _OPENMP
    nthr = prm->n_threads > 0 ? prm->n_threads : omp_get_max_threads();
#endif

    /* convertMatricesToPosesCU, SBA.cu:71-79 (all images valid, SBA.cpp:99-101) */
    for (int k = 0; k < N; k++) orc_matrix_to_pose(poses + 16 * k, xRot + 3 * k, xTrans + 3 * k);

    /* ORC_KEEP_T=1 (tests/tools/reference_order_experiment.py only): the first iterate's T is the INPUT matrix as it stands instead of Exp(Log(.)) of it, so that
     * an experiment can start this oracle and the HIP path (same switch there: BTBA_PREPARE_KEEP_T) from bit-identical matrices -- host and device libm differ
     * in the last bits of sinf / cosf / asinf.  Not the reference's behaviour; never set by the tests. */
    const int keep_T = getenv("ORC_KEEP_T") != NULL;
    for (int it = 0; it < prm->n_gn_iters; it++) {
        const float wS = prm->weight_sparse, wD = prm->weight_dense_depth;   /* constant per iteration, SBA.cpp:27-32 */
        int use_dense = (wD > 0);
        /* convertLiePosesToMatricesCU_Kernel, SolverBundling.cu:890-897 */
        for (int k = 0; k < N; k++) {
            if (it == 0 && keep_T) memcpy(T + 16 * k, poses + 16 * k, 16 * sizeof(float));
            else orc_pose_to_matrix(xRot + 3 * k, xTrans + 3 * k, T + 16 * k);
            m4_inverse(T + 16 * k, Tinv + 16 * k);
        }
        if (use_dense) {
            if (P == 0) use_dense = 0;                           /* "no overlapping images", :280-283 */
            else build_dense_system(prm, N, Wd, Hd, intr, campos, normals, pairs, P, T, Tinv, JtJ, Jtr, cnts);
        }
        if (tr && tr->dense_JtJ) { if (use_dense) memcpy(tr->dense_JtJ + (size_t)it * dim * dim, JtJ, sizeof(float) * dim * dim); else memset(tr->dense_JtJ + (size_t)it * dim * dim, 0, sizeof(float) * dim * dim); }
        if (tr && tr->dense_Jtr) { if (use_dense) memcpy(tr->dense_Jtr + (size_t)it * dim, Jtr, sizeof(float) * dim); else memset(tr->dense_Jtr + (size_t)it * dim, 0, sizeof(float) * dim); }
        if (tr && tr->dense_count) for (int p = 0; p < P; p++) tr->dense_count[(size_t)it * P + p] = use_dense ? cnts[p] : 0;

        /* Initialization: PCGInit_Kernel1/2, SolverBundling.cu:575-651 */
        float rz = 0.0f;
        {
            double acc = 0.0; float accf = 0.0f;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthr)
            for (int k = 1; k < N; k++) {
                dRot[3 * k] = dRot[3 * k + 1] = dRot[3 * k + 2] = 0.0f; dTrans[3 * k] = dTrans[3 * k + 1] = dTrans[3 * k + 2] = 0.0f;
                eval_minus_JTF(prm, k, corr, &tab, T, use_dense, Jtr, rRot + 3 * k, rTrans + 3 * k, mRot + 3 * k, mTrans + 3 * k);
                for (int q = 0; q < 3; q++) { pRot[3 * k + q] = mRot[3 * k + q] * rRot[3 * k + q]; pTrans[3 * k + q] = mTrans[3 * k + q] * rTrans[3 * k + q]; }
                for (int q = 0; q < 3; q++) { ApRot[3 * k + q] = 0.0f; ApTrans[3 * k + q] = 0.0f; }
            }
            for (int k = 1; k < N; k++) {
                float d = dot3(rRot + 3 * k, pRot + 3 * k) + dot3(rTrans + 3 * k, pTrans + 3 * k);
                acc += d; accf += d;
            }
            rz = prm->accum_mode ? (float)acc : accf;
        }
        if (tr && tr->rhs) for (int k = 0; k < N; k++) for (int q = 0; q < 3; q++) { tr->rhs[((size_t)it * N + k) * 6 + q] = k ? rRot[3 * k + q] : 0.0f; tr->rhs[((size_t)it * N + k) * 6 + 3 + q] = k ? rTrans[3 * k + q] : 0.0f; }
        if (tr && tr->precond) for (int k = 0; k < N; k++) for (int q = 0; q < 3; q++) { tr->precond[((size_t)it * N + k) * 6 + q] = k ? mRot[3 * k + q] : 0.0f; tr->precond[((size_t)it * N + k) * 6 + 3 + q] = k ? mTrans[3 * k + q] : 0.0f; }

        /* PCGIteration<useSparse,useDense>, SolverBundling.cu:820-887 */
        for (int li = 0; li < prm->n_pcg_iters; li++) {
            if (wS > 0.0f) {
                /* PCGStep_Kernel0 :692-702 */
#pragma omp parallel for schedule(static) num_threads(nthr)
                for (int x = 0; x < C; x++) apply_J(prm, &corr[x], T, pRot, pTrans, Jp + 3 * (size_t)x);
                /* PCGStep_Kernel1a :704-726 */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthr)
                for (int k = 1; k < N; k++) {
                    float oR[3], oT[3];
                    apply_JT(prm, k, corr, &tab, T, Jp, oR, oT);
                    for (int q = 0; q < 3; q++) { ApRot[3 * k + q] += oR[q]; ApTrans[3 * k + q] += oT[q]; }
                }
            }
            if (use_dense) {
                /* PCGStep_Kernel_Dense :668-690 */
                for (int k = 1; k < N; k++) {
                    float oR[3], oT[3];
                    apply_JTJ_dense(k, N, JtJ, pRot, pTrans, oR, oT);
                    for (int q = 0; q < 3; q++) { ApRot[3 * k + q] += oR[q]; ApTrans[3 * k + q] += oT[q]; }
                }
            }
            /* PCGStep_Kernel1b :728-744 */
            double pAp_d = 0.0; float pAp_f = 0.0f;
            for (int k = 1; k < N; k++) { float d = dot3(pRot + 3 * k, ApRot + 3 * k) + dot3(pTrans + 3 * k, ApTrans + 3 * k); pAp_d += d; pAp_f += d; }
            float pAp = prm->accum_mode ? (float)pAp_d : pAp_f;
            /* PCGStep_Kernel2 :746-781 */
            float alpha = 0.0f;
            if (pAp > 0.000001f) alpha = rz / pAp;
            double b_d = 0.0; float b_f = 0.0f;
            for (int k = 1; k < N; k++) {
                for (int q = 0; q < 3; q++) {
                    dRot[3 * k + q] = dRot[3 * k + q] + alpha * pRot[3 * k + q];
                    dTrans[3 * k + q] = dTrans[3 * k + q] + alpha * pTrans[3 * k + q];
                    rRot[3 * k + q] = rRot[3 * k + q] - alpha * ApRot[3 * k + q];
                    rTrans[3 * k + q] = rTrans[3 * k + q] - alpha * ApTrans[3 * k + q];
                    zRot[3 * k + q] = mRot[3 * k + q] * rRot[3 * k + q];
                    zTrans[3 * k + q] = mTrans[3 * k + q] * rTrans[3 * k + q];
                }
                float d = dot3(zRot + 3 * k, rRot + 3 * k) + dot3(zTrans + 3 * k, rTrans + 3 * k);
                b_d += d; b_f += d;
            }
            float rz_new = prm->accum_mode ? (float)b_d : b_f;
            /* PCGStep_Kernel3 :783-818 */
            float beta = 0.0f;
            if (rz > 0.000001f) beta = rz_new / rz;
            if (tr && tr->pcg_scalars) { float *s = tr->pcg_scalars + ((size_t)it * prm->n_pcg_iters + li) * 4; s[0] = pAp; s[1] = alpha; s[2] = rz_new; s[3] = beta; }
            rz = rz_new;
            for (int k = 1; k < N; k++)
                for (int q = 0; q < 3; q++) {
                    pRot[3 * k + q] = zRot[3 * k + q] + beta * pRot[3 * k + q];
                    pTrans[3 * k + q] = zTrans[3 * k + q] + beta * pTrans[3 * k + q];
                    ApRot[3 * k + q] = 0.0f; ApTrans[3 * k + q] = 0.0f;
                }
            if (li == prm->n_pcg_iters - 1) {
                for (int k = 1; k < N; k++) {
                    float nW[3], nT[3];
                    lie_update(dRot + 3 * k, dTrans + 3 * k, xRot + 3 * k, xTrans + 3 * k, nW, nT);
                    for (int q = 0; q < 3; q++) { xRot[3 * k + q] = nW[q]; xTrans[3 * k + q] = nT[q]; }
                }
            }
        }
        if (tr && tr->delta) for (int k = 0; k < N; k++) for (int q = 0; q < 3; q++) { tr->delta[((size_t)it * N + k) * 6 + q] = k ? dRot[3 * k + q] : 0.0f; tr->delta[((size_t)it * N + k) * 6 + 3 + q] = k ? dTrans[3 * k + q] : 0.0f; }
        if (tr && tr->x_after) for (int k = 0; k < N; k++) for (int q = 0; q < 3; q++) { tr->x_after[((size_t)it * N + k) * 6 + q] = xRot[3 * k + q]; tr->x_after[((size_t)it * N + k) * 6 + 3 + q] = xTrans[3 * k + q]; }
        if (tr && tr->T_after) for (int k = 0; k < N; k++) orc_pose_to_matrix(xRot + 3 * k, xTrans + 3 * k, tr->T_after + ((size_t)it * N + k) * 16);
    }
    /* convertPosesToMatricesCU, SBA.cu:97-104 */
    for (int k = 0; k < N; k++) orc_pose_to_matrix(xRot + 3 * k, xTrans + 3 * k, poses + 16 * k);

    free(corr); free(tab.rows); free(tab.counts);
    free(xRot); free(xTrans); free(T); free(Tinv); free(JtJ); free(Jtr);
    free(rRot); free(rTrans); free(zRot); free(zTrans); free(pRot); free(pTrans); free(dRot); free(dTrans);
    free(ApRot); free(ApTrans); free(mRot); free(mTrans); free(Jp); free(cnts);
    return 0;
}

/* ---- test hooks (one reference function each) ------------------------------------------- */
ORC_API void orc_lie_deriv_I(const float *A, const float *D, const float *p, float *jac) { lie_deriv_I(A, D, p, jac); }
ORC_API void orc_lie_deriv_J(const float *A, const float *D, const float *p, float *jac) { lie_deriv_J(A, D, p, jac); }
ORC_API int orc_bilinear4(float x, float y, const float *img, int W, int H, float *out) { return bilinear4(x, y, img, (unsigned)W, (unsigned)H, out); }
ORC_API float orc_huber_weight(float e, float delta) { return huber_weight(e, delta); }
ORC_API void orc_lie_update(const float *dW, const float *dT, const float *cW, const float *cT, float *nW, float *nT) { lie_update(dW, dT, cW, cT, nW, nT); }
ORC_API void orc_params_default(orc_params *p)
{
    p->n_gn_iters = 7; p->n_pcg_iters = 5; p->robust_delta = 0.005f;
    p->dense_dist_thresh = 0.02f; p->dense_normal_thresh = (float)cos(45.0 / 180.0 * M_PI);
    p->depth_min = 0.1f; p->depth_max = 9999.0f;
    p->weight_sparse = 1.0f; p->weight_dense_depth = 1.0f;
    p->accum_mode = 1; p->n_threads = 0;
}
/* The sparse operator alone (matrix-free J^T J p as the reference applies it) for tests:
 * out[N][6] = (ApRot, ApTrans) given p[N][6] = (pRot, pTrans) and matrices T[N][16]. */
ORC_API int orc_sparse_apply(const orc_params *prm, int N, const orc_entryj *corr_in, int C, const float *T, const float *p, float *out)
{
    orc_entryj *corr = (orc_entryj *)malloc(sizeof(orc_entryj) * (C > 0 ? C : 1));
    if (C > 0) memcpy(corr, corr_in, sizeof(orc_entryj) * C);
    corr_table tab;
    build_table(corr, C, N, 2 * C + 1, &tab);
    float *pR = (float *)calloc(3 * N, 4), *pT = (float *)calloc(3 * N, 4), *Jp = (float *)calloc(3 * (size_t)(C > 0 ? C : 1), 4);
    for (int k = 0; k < N; k++) for (int q = 0; q < 3; q++) { pR[3 * k + q] = p[6 * k + q]; pT[3 * k + q] = p[6 * k + 3 + q]; }
    for (int x = 0; x < C; x++) apply_J(prm, &corr[x], T, pR, pT, Jp + 3 * (size_t)x);
    for (int k = 0; k < N; k++) {
        float oR[3] = { 0, 0, 0 }, oT[3] = { 0, 0, 0 };
        if (k > 0) apply_JT(prm, k, corr, &tab, T, Jp, oR, oT);
        for (int q = 0; q < 3; q++) { out[6 * k + q] = oR[q]; out[6 * k + 3 + q] = oT[q]; }
    }
    free(corr); free(tab.rows); free(tab.counts); free(pR); free(pT); free(Jp);
    return 0;
}

/* ==== SURVEY.md 8(f) rank 3: depth pre-processing and normals (Frame::processDepth, Frame.cpp:152-180;
 * Frame::depthToCloudAndNormals, Frame.cpp:182-233) ========================================= */

/* erodeDepthMapDevice, CUDAImageUtil.cu:676-718.  Out-of-image neighbours are not counted but the
 * denominator is the full window. */
ORC_API void orc_erode_depth(const float *in, float *out, int W, int H, int radius, float dThresh, float fracReq)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            unsigned count = 0;
            const float oldDepth = in[y * W + x];
            if (oldDepth <= 0.1f) { out[y * W + x] = 0; continue; }
            for (int i = -radius; i <= radius; i++)
                for (int j = -radius; j <= radius; j++)
                    if (x + j >= 0 && x + j < W && y + i >= 0 && y + i < H) {
                        const float d = in[(y + i) * W + (x + j)];
                        if (d == -INFINITY || d < 0.1f || fabsf(d - oldDepth) > dThresh) count++;
                    }
            const unsigned sum = (2 * radius + 1) * (2 * radius + 1);
            out[y * W + x] = ((float)count / (float)sum >= fracReq) ? 0.0f : in[y * W + x];
        }
}

/* gaussFilterDepthMapDevice, CUDAImageUtil.cu:735-797 (a mean-gated bilateral filter) */
ORC_API void orc_gauss_filter_depth(const float *in, float *out, int W, int H, int radius, float sigmaD, float sigmaR)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            out[y * W + x] = 0;
            const float depthCenter = in[y * W + x];
            float mean_depth = 0;
            int num_valid = 0;
            for (int m = x - radius; m <= x + radius; m++)
                for (int n = y - radius; n <= y + radius; n++)
                    if (m >= 0 && n >= 0 && m < W && n < H) {
                        const float c = in[n * W + m];
                        if (c >= 0.1f) { num_valid++; mean_depth += c; }
                    }
            if (num_valid == 0) continue;
            mean_depth /= num_valid;
            float sum = 0.0f, sumWeight = 0.0f;
            for (int m = x - radius; m <= x + radius; m++)
                for (int n = y - radius; n <= y + radius; n++)
                    if (m >= 0 && n >= 0 && m < W && n < H) {
                        const float c = in[n * W + m];
                        if (c >= 0.1f && (double)fabsf(c - mean_depth) < 0.01) {
                            const float weight = expf(-((m - x) * (m - x) + (y - n) * (y - n)) / (2.0f * sigmaD * sigmaD) - (depthCenter - c) * (depthCenter - c) / (2 * sigmaR * sigmaR));
                            sumWeight += weight;
                            sum += weight * c;
                        }
                    }
            const float num_total = (float)((2 * radius + 1) * (2 * radius + 1));
            if (sumWeight > 0.0f && num_valid / num_total > 0) out[y * W + x] = sum / sumWeight;
        }
}

/* Frame::processDepth, Frame.cpp:152-180: erode, then the filter twice */
ORC_API void orc_process_depth(const float *in, float *out, int W, int H, int erode_radius, float erode_diff, float erode_ratio,
                               int bf_radius, float sigmaD, float sigmaR)
{
    float *a = (float *)malloc(sizeof(float) * (size_t)W * H), *b = (float *)malloc(sizeof(float) * (size_t)W * H);
    orc_erode_depth(in, a, W, H, erode_radius, erode_diff, erode_ratio);
    orc_gauss_filter_depth(a, b, W, H, bf_radius, sigmaD, sigmaR);
    orc_gauss_filter_depth(b, out, W, H, bf_radius, sigmaD, sigmaR);
    free(a); free(b);
}

/* convertDepthFloatToCameraSpaceFloat4 (CUDAImageUtil.cu:310-327) with K^-1 embedded in a 4x4 whose last row
 * is (0,0,0,1) (Frame.cpp:187-197), then computeNormals_Kernel (CUDAImageUtil.cu:342-412).
 * xyz may be NULL.  Kinv: row-major 4x4 (the generic cofactor inverse of the embedding, like the device path). */
ORC_API void orc_depth_to_normals(const float *depth, int W, int H, const float *Kinv, float *normals, float *xyz_out)
{
    float *xyz = (float *)calloc((size_t)W * H * 4, sizeof(float));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float d = depth[y * W + x];
            float *o = xyz + 4 * ((size_t)y * W + x);
            if (d >= 0.1) {
                const float v[4] = { (float)x * d, (float)y * d, d, d };
                float c[4];
                m4_vec4(Kinv, v, c);
                o[0] = c[0]; o[1] = c[1]; o[2] = c[3]; o[3] = 1.0f;       /* (x, y, cameraSpace.w, 1) */
            }
        }
    const float z_diff_thres = 0.02f;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float *o = normals + 4 * ((size_t)y * W + x);
            o[0] = o[1] = o[2] = o[3] = 0;
            if (!(x > 0 && x < W - 1 && y > 0 && y < H - 1)) continue;
            const float *CC = xyz + 4 * ((size_t)y * W + x), *PC = xyz + 4 * ((size_t)(y + 1) * W + x), *CP = xyz + 4 * ((size_t)y * W + x + 1);
            const float *MC = xyz + 4 * ((size_t)(y - 1) * W + x), *CM = xyz + 4 * ((size_t)y * W + x - 1);
            if (CC[2] < 0.1f) continue;
            float xd[3], yd[3];
            if (PC[2] >= 0.1f && MC[2] >= 0.1f && fabsf(PC[2] - CC[2]) <= z_diff_thres && fabsf(MC[2] - CC[2]) <= z_diff_thres) { for (int k = 0; k < 3; k++) xd[k] = PC[k] - MC[k]; }
            else if (PC[2] >= 0.1f && fabsf(PC[2] - CC[2]) <= z_diff_thres) { for (int k = 0; k < 3; k++) xd[k] = PC[k] - CC[k]; }
            else if (MC[2] >= 0.1f && fabsf(MC[2] - CC[2]) <= z_diff_thres) { for (int k = 0; k < 3; k++) xd[k] = MC[k] - CC[k]; }
            else continue;
            if (CP[2] >= 0.1f && CM[2] >= 0.1f && fabsf(CP[2] - CC[2]) <= z_diff_thres && fabsf(CM[2] - CC[2]) <= z_diff_thres) { for (int k = 0; k < 3; k++) yd[k] = CP[k] - CM[k]; }
            else if (CP[2] >= 0.1f && fabsf(CP[2] - CC[2]) <= z_diff_thres) { for (int k = 0; k < 3; k++) yd[k] = CP[k] - CC[k]; }
            else if (CM[2] >= 0.1f && fabsf(CM[2] - CC[2]) <= z_diff_thres) { for (int k = 0; k < 3; k++) yd[k] = CM[k] - CC[k]; }
            else continue;
            float n[3];
            cross3(xd, yd, n);
            const float l = len3(n);
            n[0] = n[0] / l; n[1] = n[1] / l; n[2] = n[2] / l;
            const float mcc[3] = { -CC[0], -CC[1], -CC[2] };
            if (dot3(n, mcc) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
            if (l > 0.0f) { o[0] = n[0]; o[1] = n[1]; o[2] = n[2]; o[3] = 0.0f; }
        }
    if (xyz_out) memcpy(xyz_out, xyz, sizeof(float) * 4 * (size_t)W * H);
    free(xyz);
}
