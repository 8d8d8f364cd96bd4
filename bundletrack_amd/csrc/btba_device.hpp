// btba_device.hpp -- device-side SE(3) math and wave64 reductions for gfx950.
//
// The SE(3) helpers restate (not copy) the branch structure and thresholds of the reference's
//   src/cuda/Solver/LieDerivUtil.h:17-201 (exp_rotation, ln_rotation, matrixToPose, poseToMatrix,
//   computeLieUpdate) and the generic cofactor inverse cuda_SimpleMatrixUtil.h:978-1104,
// because the solver's iterates depend on exactly those thresholds (theta^2 < 1e-8 / 1e-6,
// cos > 0.7071.., theta > 1e-5 / 1e-3).  Matrices are row-major float[16] in registers.
#pragma once
#include <hip/hip_runtime.h>

namespace btba {

struct Mat4 { float m[16]; };

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) { return ax * bx + ay * by + az * bz; }

__device__ __forceinline__ Mat4 mat_mul(const Mat4 &a, const Mat4 &b)
{
    Mat4 o;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            o.m[4 * r + c] = a.m[4 * r] * b.m[c] + a.m[4 * r + 1] * b.m[4 + c] + a.m[4 * r + 2] * b.m[8 + c] + a.m[4 * r + 3] * b.m[12 + c];
    return o;
}

// rigid transform of a point (w = 1), float4x4::operator*(float3)
__device__ __forceinline__ void xform_point(const Mat4 &t, float x, float y, float z, float &ox, float &oy, float &oz)
{
    ox = t.m[0] * x + t.m[1] * y + t.m[2] * z + t.m[3];
    oy = t.m[4] * x + t.m[5] * y + t.m[6] * z + t.m[7];
    oz = t.m[8] * x + t.m[9] * y + t.m[10] * z + t.m[11];
}

__device__ __forceinline__ float minor3(const float *m, int r0, int r1, int r2, int c0, int c1, int c2)
{
    return m[4 * r0 + c0] * (m[4 * r1 + c1] * m[4 * r2 + c2] - m[4 * r1 + c2] * m[4 * r2 + c1])
         - m[4 * r0 + c1] * (m[4 * r1 + c0] * m[4 * r2 + c2] - m[4 * r1 + c2] * m[4 * r2 + c0])
         + m[4 * r0 + c2] * (m[4 * r1 + c0] * m[4 * r2 + c1] - m[4 * r1 + c1] * m[4 * r2 + c0]);
}

// generic 4x4 inverse by cofactors (the reference does NOT use the rigid shortcut)
// float4x4::getInverse (cuda_SimpleMatrixUtil.h:978-1104): adjugate entry (R, C) = cofactor of element (C, R), six signed
// triple products summed left to right in the reference's order (see oracle/btba_oracle.c m4_inverse), scaled by 1/det.
__device__ __forceinline__ Mat4 mat_inverse(const Mat4 &a)
{
    Mat4 adj;
#pragma unroll
    for (int R = 0; R < 4; R++) {
#pragma unroll
        for (int Cc = 0; Cc < 4; Cc++) {
            const int r0 = (Cc == 0) ? 1 : 0, r1 = (Cc <= 1) ? 2 : 1, r2 = (Cc <= 2) ? 3 : 2;
            const int c0 = (R == 0) ? 1 : 0, c1 = (R <= 1) ? 2 : 1, c2 = (R <= 2) ? 3 : 2;
            const float t1 = a.m[4 * r0 + c0] * a.m[4 * r1 + c1] * a.m[4 * r2 + c2], t2 = a.m[4 * r0 + c0] * a.m[4 * r1 + c2] * a.m[4 * r2 + c1];
            const float t3 = a.m[4 * r1 + c0] * a.m[4 * r0 + c1] * a.m[4 * r2 + c2], t4 = a.m[4 * r1 + c0] * a.m[4 * r0 + c2] * a.m[4 * r2 + c1];
            const float t5 = a.m[4 * r2 + c0] * a.m[4 * r0 + c1] * a.m[4 * r1 + c2], t6 = a.m[4 * r2 + c0] * a.m[4 * r0 + c2] * a.m[4 * r1 + c1];
            adj.m[4 * R + Cc] = ((R + Cc) & 1) ? (((((-t1) + t2) + t3) - t4) - t5) + t6 : ((((t1 - t2) - t3) + t4) + t5) - t6;
        }
    }
    float det = a.m[0] * adj.m[0] + a.m[1] * adj.m[4] + a.m[2] * adj.m[8] + a.m[3] * adj.m[12];
    float rdet = 1.0f / det;
    Mat4 o;
#pragma unroll
    for (int k = 0; k < 16; k++) o.m[k] = adj.m[k] * rdet;
    return o;
}

// Division and square root of the SE(3) helpers.  FAST = false: IEEE (v_div_scale / v_div_fmas / v_div_fixup sequences, refined square root) -- what the seam
// kernels (k_prepare, k_poses_to_matrices) and k_system_solve use.  FAST = true (round 6, the update phase of k_solve_small / k_solve_mid): x * v_rcp_f32(y) and
// v_sqrt_f32, 1 ulp each -- the class of arithmetic the reference's own build has (-use_fast_math: --prec-div=false --prec-sqrt=false, CMakeLists.txt:7) and the
// PCG's alpha / beta already use; ten divisions and nine square roots are ~170 of the ~650 dependent instructions of the one-lane-per-frame update chain.
template <bool FAST> __device__ __forceinline__ float se3_div(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }
template <bool FAST> __device__ __forceinline__ float se3_sqrt(float x) { return FAST ? __builtin_amdgcn_sqrtf(x) : sqrtf(x); }

#define BTBA_ONE_TWENTIETH 0.05f
#define BTBA_ONE_SIXTH 0.16666667f

// R (row-major 3x3 into r[9]) from w with coefficients A, B  (rodrigues_so3_exp)
__device__ __forceinline__ void rodrigues(const float w[3], float A, float B, float r[9])
{
    const float wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
    r[0] = 1.0f - B * (wy2 + wz2);
    r[4] = 1.0f - B * (wx2 + wz2);
    r[8] = 1.0f - B * (wx2 + wy2);
    float a = A * w[2], b = B * (w[0] * w[1]);
    r[1] = b - a; r[3] = b + a;
    a = A * w[1]; b = B * (w[0] * w[2]);
    r[2] = b + a; r[6] = b - a;
    a = A * w[0]; b = B * (w[1] * w[2]);
    r[5] = b - a; r[7] = b + a;
}

template <bool FAST = false>
__device__ __forceinline__ void exp_rotation(const float w[3], float r[9])
{
    const float theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const float theta = se3_sqrt<FAST>(theta_sq);
    float A, B;
    if ((double)theta_sq < 1e-8) {
        A = 1.0f - BTBA_ONE_SIXTH * theta_sq;
        B = 0.5f;
    } else if ((double)theta_sq < 1e-6) {
        B = 0.5f - 0.25f * BTBA_ONE_SIXTH * theta_sq;
        A = 1.0f - theta_sq * BTBA_ONE_SIXTH * (1.0f - BTBA_ONE_TWENTIETH * theta_sq);
    } else {
        const float inv_theta = se3_div<FAST>(1.0f, theta);
        float sn, cs;
        sincosf(theta, &sn, &cs);                     // one argument reduction for both (the values of sinf / cosf, bit for bit)
        A = sn * inv_theta;
        B = (1.0f - cs) * (inv_theta * inv_theta);
    }
    rodrigues(w, A, B, r);
}

template <bool FAST = false>
__device__ __forceinline__ void ln_rotation(const float R[9], float out[3])
{
    const float cos_angle = ((R[0] + R[4] + R[8]) - 1.0f) * 0.5f;
    float r0 = (R[7] - R[5]) * 0.5f, r1 = (R[2] - R[6]) * 0.5f, r2 = (R[3] - R[1]) * 0.5f;
    const float sin_angle_abs = se3_sqrt<FAST>(r0 * r0 + r1 * r1 + r2 * r2);
    if (cos_angle > 0.70710678118654752440f) {
        if (sin_angle_abs > 0) {
            const float s = se3_div<FAST>(asinf(sin_angle_abs), sin_angle_abs);
            r0 *= s; r1 *= s; r2 *= s;
        }
    } else if (cos_angle > -0.70710678118654752440f) {
        const float s = se3_div<FAST>(acosf(cos_angle), sin_angle_abs);
        r0 *= s; r1 *= s; r2 *= s;
    } else {
        const float angle = 3.14159265358979323846f - asinf(sin_angle_abs);
        const float d0 = R[0] - cos_angle, d1 = R[4] - cos_angle, d2 = R[8] - cos_angle;
        float q0, q1, q2;
        if (fabsf(d0) > fabsf(d1) && fabsf(d0) > fabsf(d2)) {
            q0 = d0; q1 = (R[3] + R[1]) * 0.5f; q2 = (R[2] + R[6]) * 0.5f;
        } else if (fabsf(d1) > fabsf(d2)) {
            q0 = (R[3] + R[1]) * 0.5f; q1 = d1; q2 = (R[7] + R[5]) * 0.5f;
        } else {
            q0 = (R[2] + R[6]) * 0.5f; q1 = (R[7] + R[5]) * 0.5f; q2 = d2;
        }
        if (q0 * r0 + q1 * r1 + q2 * r2 < 0) { q0 = -q0; q1 = -q1; q2 = -q2; }
        const float s = se3_div<FAST>(angle, se3_sqrt<FAST>(q0 * q0 + q1 * q1 + q2 * q2));
        r0 = q0 * s; r1 = q1 * s; r2 = q2 * s;
    }
    out[0] = r0; out[1] = r1; out[2] = r2;
}

// SE(3) log: 4x4 -> (rot, trans)   (matrixToPose)
// The reference evaluates sinf(theta / 2) for the translation's scale and, inside exp_rotation(-rot / 2), sinf and cosf of |rot / 2| again:
// |-rot / 2| = sqrtf(sum (rot_i / 2)^2) = theta / 2 BIT FOR BIT (scaling by a power of two commutes with every rounding of the sum and of the
// correctly rounded square root), so one sincosf serves both -- ~40 instructions off k_solve_small's one-lane update chain, same bits
// (tests/test_gpu_parity.py::test_se3_helpers_bit_exact against the reference's own LieDerivUtil.h).
template <bool FAST = false>
__device__ __forceinline__ void matrix_to_pose(const Mat4 &M, float rot[3], float trans[3])
{
    const float R[9] = { M.m[0], M.m[1], M.m[2], M.m[4], M.m[5], M.m[6], M.m[8], M.m[9], M.m[10] };
    const float t[3] = { M.m[3], M.m[7], M.m[11] };
    ln_rotation<FAST>(R, rot);
    const float theta = se3_sqrt<FAST>(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]);
    float shtot = 0.5f, sn = 0.0f, cs = 1.0f;
    if (theta > 0.00001f) { sincosf(theta * 0.5f, &sn, &cs); shtot = se3_div<FAST>(sn, theta); }
    const float rh[3] = { rot[0] * -0.5f, rot[1] * -0.5f, rot[2] * -0.5f };
    float Hh[9];
    {   // exp_rotation(rh, Hh) with the sine and cosine of |rh| = theta / 2 from above
        const float theta_sq = rh[0] * rh[0] + rh[1] * rh[1] + rh[2] * rh[2];
        const float theta_h = se3_sqrt<FAST>(theta_sq);
        float A, B;
        if ((double)theta_sq < 1e-8) {
            A = 1.0f - BTBA_ONE_SIXTH * theta_sq;
            B = 0.5f;
        } else if ((double)theta_sq < 1e-6) {
            B = 0.5f - 0.25f * BTBA_ONE_SIXTH * theta_sq;
            A = 1.0f - theta_sq * BTBA_ONE_SIXTH * (1.0f - BTBA_ONE_TWENTIETH * theta_sq);
        } else {
            const float inv_theta = se3_div<FAST>(1.0f, theta_h);
            A = sn * inv_theta;
            B = (1.0f - cs) * (inv_theta * inv_theta);
        }
        rodrigues(rh, A, B, Hh);
    }
    float tr0 = Hh[0] * t[0] + Hh[1] * t[1] + Hh[2] * t[2];
    float tr1 = Hh[3] * t[0] + Hh[4] * t[1] + Hh[5] * t[2];
    float tr2 = Hh[6] * t[0] + Hh[7] * t[1] + Hh[8] * t[2];
    const float tdr = t[0] * rot[0] + t[1] * rot[1] + t[2] * rot[2];
    float s;
    if (theta > 0.001f) s = se3_div<FAST>(tdr * (1.0f - 2.0f * shtot), rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]);
    else s = se3_div<FAST>(tdr, 24.0f);
    tr0 -= rot[0] * s; tr1 -= rot[1] * s; tr2 -= rot[2] * s;
    const float k = se3_div<FAST>(1.0f, 2.0f * shtot);
    trans[0] = tr0 * k; trans[1] = tr1 * k; trans[2] = tr2 * k;
}

// SE(3) exp: (rot, trans) -> 4x4   (poseToMatrix)
template <bool FAST = false>
__device__ __forceinline__ Mat4 pose_to_matrix(const float rot[3], const float trans[3])
{
    const float theta_sq = rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2];
    const float theta = se3_sqrt<FAST>(theta_sq);
    float A, B;
    const float cr[3] = { rot[1] * trans[2] - rot[2] * trans[1], rot[2] * trans[0] - rot[0] * trans[2], rot[0] * trans[1] - rot[1] * trans[0] };
    float tx, ty, tz;
    if ((double)theta_sq < 1e-8) {
        A = 1.0f - BTBA_ONE_SIXTH * theta_sq;
        B = 0.5f;
        tx = trans[0] + 0.5f * cr[0]; ty = trans[1] + 0.5f * cr[1]; tz = trans[2] + 0.5f * cr[2];
    } else {
        float C;
        if ((double)theta_sq < 1e-6) {
            C = BTBA_ONE_SIXTH * (1.0f - BTBA_ONE_TWENTIETH * theta_sq);
            A = 1.0f - theta_sq * C;
            B = 0.5f - 0.25f * BTBA_ONE_SIXTH * theta_sq;
        } else {
            const float inv_theta = se3_div<FAST>(1.0f, theta);
            float sn, cs;
            sincosf(theta, &sn, &cs);
            A = sn * inv_theta;
            B = (1.0f - cs) * (inv_theta * inv_theta);
            C = (1.0f - A) * (inv_theta * inv_theta);
        }
        const float wc[3] = { rot[1] * cr[2] - rot[2] * cr[1], rot[2] * cr[0] - rot[0] * cr[2], rot[0] * cr[1] - rot[1] * cr[0] };
        tx = trans[0] + B * cr[0] + C * wc[0]; ty = trans[1] + B * cr[1] + C * wc[1]; tz = trans[2] + B * cr[2] + C * wc[2];
    }
    float R[9];
    rodrigues(rot, A, B, R);
    Mat4 M;
    M.m[0] = R[0]; M.m[1] = R[1]; M.m[2] = R[2];  M.m[3] = tx;
    M.m[4] = R[3]; M.m[5] = R[4]; M.m[6] = R[5];  M.m[7] = ty;
    M.m[8] = R[6]; M.m[9] = R[7]; M.m[10] = R[8]; M.m[11] = tz;
    M.m[12] = 0.0f; M.m[13] = 0.0f; M.m[14] = 0.0f; M.m[15] = 1.0f;
    return M;
}

// Huber IRLS weight rho' (huberLoss .y, SolverBundlingUtil.h:24-40: float sqrt held in a double)
__device__ __forceinline__ float huber_weight(float e, float delta)
{
    const float dsqr = delta * delta;
    if (e <= dsqr) return 1.0f;
    const double sq = (double)sqrtf(e);
    return (float)((double)delta / sq);
}

// A pointer to data that no kernel launch writes while it reads it (poses, offsets, descriptors of the previous launch), as a CONSTANT
// address-space pointer: a wave-uniform load through it becomes a scalar load (s_load_dword*) whatever else the kernel stores --
// inside the persistent sweep the compiler otherwise assumes the previous item's stores may alias and keeps such values in vector registers.
template <class T> __device__ __forceinline__ const __attribute__((address_space(4))) T *as_const(const T *p) { return (const __attribute__((address_space(4))) T *)p; }

typedef float btba_f4v __attribute__((ext_vector_type(4)));
typedef int btba_i4v __attribute__((ext_vector_type(4)));
// 16 aligned bytes through a constant-address-space pointer (the HIP vector classes cannot be copied out of another address space)
__device__ __forceinline__ float4 ld_const_f4(const void *p) { const btba_f4v v = *as_const(reinterpret_cast<const btba_f4v *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
// read-once streams (a sweep's correspondences): optionally non-temporal loads, so that they do not evict the frames the dense taps keep hitting in L2
typedef float btba_f2v __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ float2 ld_stream_f2(const float2 *p)
{
    if (!NT) return *p;
    const btba_f2v v = __builtin_nontemporal_load(reinterpret_cast<const btba_f2v *>(p));
    return make_float2(v.x, v.y);
}
template <bool NT> __device__ __forceinline__ float4 ld_stream_f4(const float4 *p)
{
    if (!NT) return *p;
    const btba_f4v v = __builtin_nontemporal_load(reinterpret_cast<const btba_f4v *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int4 ld_const_i4(const void *p) { const btba_i4v v = *as_const(reinterpret_cast<const btba_i4v *>(p)); return make_int4(v.x, v.y, v.z, v.w); }

// threadIdx.x through an opaque copy.  Inside the item loop of the persistent sweep (k_fused_sweeps) everything derived from the thread
// index is loop-invariant to the compiler, which hoists it all out of the loop and then spills it (52 vector spills); taken through this
// function at the top of each per-item block, the values live only as long as the item.
__device__ __forceinline__ unsigned item_tid() { unsigned t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }

// ---- wave64 reductions on DPP (no LDS, no ds_bpermute) -----------------------------------
// Butterfly inside each row of 16 lanes (quad_perm xor1, xor2, row_half_mirror, row_mirror: every
// lane of the row ends with the row sum), then row_bcast:15 / row_bcast:31 fold the four rows so
// that lane 63 holds the wave total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}

__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);   // row_mirror
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ float wave_sum_all(float v)
{
    v = wave_sum_to_lane63(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Wave sums of NV per-lane values at once, NV % 4 == 0, by FOLDING instead of NV separate 6-step butterflies: gfx950's
// v_permlane32_swap exchanges the upper 32 lanes of one register with the lower 32 of another, so one swap + one add folds the
// two halves of TWO values (k in the lower half of the result, k + NV/2 in the upper half); v_permlane16_swap does the same one
// level down (rows of 16 lanes), leaving NV/4 registers whose row r holds partial sums of value k + r NV/4; a 4-step DPP
// butterfly inside the rows finishes them.  NV/2 + NV/4 swaps + NV adds + NV DPP adds instead of 6 NV DPP adds (the dense
// sweep's epilogue: 28 values, 70 instructions instead of 168).  Every lane of row r of q[k] ends with the sum of value k + r NV/4.
template <int NV>
__device__ __forceinline__ void wave_fold_sums(const float (&acc)[NV], float (&q)[NV / 4])
{
    static_assert(NV % 4 == 0, "wave_fold_sums folds twice");
    constexpr int H = NV / 2, Q = NV / 4;
    float h[H];
#pragma unroll
    for (int k = 0; k < H; k++) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[k]), __float_as_uint(acc[k + H]), false, false);
        h[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int k = 0; k < Q; k++) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[k]), __float_as_uint(h[k + Q]), false, false);
        float v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
        v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
        v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
        v = dpp_add<0x140, 0xf>(v);   // row_mirror
        q[k] = v;
    }
}

// the folded sums of one wave -> dst[0 .. NV)
template <int NV>
__device__ __forceinline__ void wave_fold_store(const float (&acc)[NV], float *dst)
{
    float q[NV / 4];
    wave_fold_sums<NV>(acc, q);
    const int lane = item_tid() & 63;
    if ((lane & 15) == 0) {
#pragma unroll
        for (int k = 0; k < NV / 4; k++) dst[k + (NV / 4) * (lane >> 4)] = q[k];
    }
}

// Reduce NV per-thread registers over a workgroup of NWAVES waves into out[0..NV) (global or LDS).
// lds_scratch must hold NWAVES*NV floats.  Deterministic: fixed tree inside the wave, fixed
// wave order across waves.
// atomic: add the workgroup's sums into out[] with hardware float atomics (global_atomic_add_f32) instead of storing them --
// BTBA_REDUCE_ATOMIC, the reference's own way of summing (SolverBundlingDenseUtil.h:217-285): the order in which workgroups land is not fixed.
// mode: 0 store, 1 float atomics, 2 write-through store at agent scope (the record is read by another workgroup of the same launch: k_chain)
template <int NV, int NWAVES>
__device__ __forceinline__ void block_reduce_store(float (&acc)[NV], float *lds_scratch, float *out, int mode = 0)
{
    const int tid = (int)item_tid(), wave = tid >> 6;
    wave_fold_store<NV>(acc, lds_scratch + wave * NV);
    __syncthreads();
    for (int k = tid; k < NV; k += 64 * NWAVES) {
        float s = lds_scratch[k];
#pragma unroll
        for (int w = 1; w < NWAVES; w++) s += lds_scratch[w * NV + k];
        if (mode == 1) unsafeAtomicAdd(out + k, s);
        else if (mode == 2) __hip_atomic_store(out + k, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[k] = s;
    }
}

}  // namespace btba
