// oracle/ref_driver.cpp -- calls the REFERENCE'S OWN device functions on the CPU, to pin oracle/btba_oracle.c.
//
// TEST INFRASTRUCTURE ONLY.  The reference's solver is CUDA (__global__ kernels with block-wide reductions, atomics,
// warp shuffles: not buildable here), but the mathematics those kernels call lives in header-only `__device__`
// functions.  With oracle/ref_shim/cuda_runtime.h standing in for the CUDA built-ins they use, g++ compiles those
// headers WHERE THEY LIE under /root/reference/src/cuda (nothing is copied into this repository) and this file wraps
// them in a C ABI:
//   SE(3):   poseToMatrix, matrixToPose, computeLieUpdate, evalLie_derivI/J      Solver/LieDerivUtil.h:17-282
//            float4x4::getInverse                                                cuda_SimpleMatrixUtil.h
//   image:   bilinearInterpolationFloat4                                         Solver/ICPUtil.h:83-110
//   robust:  huberLoss                                                           Solver/SolverBundlingUtil.h:24-40
//   sparse:  evalMinusJTFDevice<false>, applyJDevice, applyJTDevice              Solver/SolverBundlingEquationsLie.h:60-211
//   dense:   findDenseCorr (float4 normals), computeJacobianBlockRow_i/j,        Solver/SolverBundlingDenseUtil.h:78-110,
//            addToLocalSystemBrute                                                 SolverBundlingEquationsLie.h:214-230, DenseUtil.h:286-314
// What stays outside: the kernels' own glue.  The few lines between those calls in BuildDenseSystem_Kernel
// (SolverBundling.cu:129-229: diff, residual, weight, which rows to build) and the frame->correspondence table
// (:1006-1031) are restated below, marked GLUE; the block reductions, PCG scalars and launch order are not
// reachable this way and remain pinned only by the self-derived tests.
// Built by `make -C oracle ref` into oracle/_ref/libbtba_ref.so when /root/reference exists.
#define _CUTIL_INLINE_H_          // cutil_inline*.h: error-check wrappers around the CUDA runtime, nothing the math needs
#define _CUTIL_H_
#include "Solver/GlobalDefines.h"
#include "cutil_math.h"
#include "cuda_SimpleMatrixUtil.h"
#include "SIFTImageManager.h"
#include "CUDACacheUtil.h"
#include "Solver/SolverBundlingState.h"
#include "Solver/SolverBundlingParameters.h"
#include "Solver/LieDerivUtil.h"
#include "Solver/ICPUtil.h"
#include "Solver/SolverBundlingUtil.h"
#include "Solver/SolverBundlingEquationsLie.h"
#include "Solver/SolverBundlingDenseUtil.h"

#include <vector>

#define REF_API extern "C" __attribute__((visibility("default")))

static float4x4 load4(const float *m) { float4x4 M; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M(r, c) = m[4 * r + c]; return M; }
static void store4(const float4x4 &M, float *m) { for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) m[4 * r + c] = M(r, c); }

REF_API void ref_pose_to_matrix(const float *rot, const float *trans, float *M)
{
    store4(poseToMatrix(make_float3(rot[0], rot[1], rot[2]), make_float3(trans[0], trans[1], trans[2])), M);
}
REF_API void ref_matrix_to_pose(const float *M, float *rot, float *trans)
{
    float3 r, t;
    matrixToPose(load4(M), r, t);
    rot[0] = r.x; rot[1] = r.y; rot[2] = r.z; trans[0] = t.x; trans[1] = t.y; trans[2] = t.z;
}
REF_API void ref_mat4_inverse(const float *M, float *out) { store4(load4(M).getInverse(), out); }
REF_API void ref_lie_update(const float *dW, const float *dT, const float *cW, const float *cT, float *nW, float *nT)
{
    float3 w, t;
    computeLieUpdate(make_float3(dW[0], dW[1], dW[2]), make_float3(dT[0], dT[1], dT[2]), make_float3(cW[0], cW[1], cW[2]), make_float3(cT[0], cT[1], cT[2]), w, t);
    nW[0] = w.x; nW[1] = w.y; nW[2] = w.z; nT[0] = t.x; nT[1] = t.y; nT[2] = t.z;
}
REF_API void ref_lie_deriv(int which_j, const float *A, const float *D, const float *p, float *jac /* 3x6 row-major */)
{
    const matNxM<3, 6> J = which_j ? evalLie_derivJ(load4(A), load4(D), make_float3(p[0], p[1], p[2])) : evalLie_derivI(load4(A), load4(D), make_float3(p[0], p[1], p[2]));
    for (int r = 0; r < 3; r++) for (int c = 0; c < 6; c++) jac[6 * r + c] = J(r, c);
}
REF_API void ref_bilinear4(float x, float y, const float *img, int W, int H, float *out)
{
    const float4 v = bilinearInterpolationFloat4(x, y, reinterpret_cast<const float4 *>(img), (unsigned)W, (unsigned)H);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
REF_API void ref_huber(float e, float delta, float *rho) { float3 r; huberLoss(e, delta, r); rho[0] = r.x; rho[1] = r.y; rho[2] = r.z; }

// ---- sparse term ---------------------------------------------------------------------------------------------
struct SparseSetup {
    SolverInput in{};
    SolverState st{};
    SolverParameters prm{};
    std::vector<EntryJ> corr;
    std::vector<int> table, counts;
    std::vector<float4x4> T;
    std::vector<float3> dRot, dTrans, pRot, pTrans, precR, precT, Jp;
    SparseSetup(int N, const float *corr_in, int C, const float *Tin, float robust_delta, float weight_sparse)
    {
        corr.resize(C);
        memcpy(corr.data(), corr_in, sizeof(EntryJ) * (size_t)C);
        // GLUE (BuildVariablesToCorrespondencesTableDevice, SolverBundling.cu:1006-1031): every valid correspondence is listed
        // under both of its frames; the reference fills the rows with atomics (arbitrary order), here in input order
        const int stride = C > 0 ? C : 1;
        table.assign((size_t)N * stride, 0); counts.assign(N, 0);
        for (int c = 0; c < C; c++) if (corr[c].isValid()) {
            table[(size_t)corr[c].imgIdx_i * stride + counts[corr[c].imgIdx_i]++] = c;
            table[(size_t)corr[c].imgIdx_j * stride + counts[corr[c].imgIdx_j]++] = c;
        }
        T.resize(N);
        for (int k = 0; k < N; k++) T[k] = load4(Tin + 16 * k);
        dRot.assign(N, make_float3(0, 0, 0)); dTrans = pRot = pTrans = precR = precT = dRot;
        Jp.assign(C > 0 ? C : 1, make_float3(0, 0, 0));
        in.d_correspondences = corr.data(); in.d_variablesToCorrespondences = table.data(); in.d_numEntriesPerRow = counts.data();
        in.numberOfCorrespondences = C; in.numberOfImages = N; in.maxNumberOfImages = N; in.maxCorrPerImage = stride;
        st.d_xTransforms = T.data(); st.d_deltaRot = dRot.data(); st.d_deltaTrans = dTrans.data();
        st.d_pRot = pRot.data(); st.d_pTrans = pTrans.data(); st.d_precondionerRot = precR.data(); st.d_precondionerTrans = precT.data(); st.d_Jp = Jp.data();
        prm.robust_delta = robust_delta; prm.weightSparse = weight_sparse; prm.useDense = false;
    }
};

// rhs (rot, trans) = -J^T W r and the Jacobi preconditioner of every frame k >= 1 (PCGInit_Kernel1 calls exactly this)
REF_API void ref_sparse_rhs(int N, const float *corr, int C, const float *T, float robust_delta, float weight_sparse, float *rhs /* [N][6] rot,trans */, float *prec /* [N][6] */)
{
    SparseSetup S(N, corr, C, T, robust_delta, weight_sparse);
    for (int k = 0; k < N; k++) {
        float3 rr = make_float3(0, 0, 0), rt = rr;
        if (k > 0) evalMinusJTFDevice<false>((unsigned)k, S.in, S.st, S.prm, rr, rt);
        rhs[6 * k] = rr.x; rhs[6 * k + 1] = rr.y; rhs[6 * k + 2] = rr.z; rhs[6 * k + 3] = rt.x; rhs[6 * k + 4] = rt.y; rhs[6 * k + 5] = rt.z;
        prec[6 * k] = S.precR[k].x; prec[6 * k + 1] = S.precR[k].y; prec[6 * k + 2] = S.precR[k].z;
        prec[6 * k + 3] = S.precT[k].x; prec[6 * k + 4] = S.precT[k].y; prec[6 * k + 5] = S.precT[k].z;
    }
}
// out = J^T (J p), the matrix-free sparse operator of one PCG step (PCGStep_Kernel0 + Kernel1a)
REF_API void ref_sparse_apply(int N, const float *corr, int C, const float *T, float weight_sparse, const float *p /* [N][6] rot,trans */, float *out)
{
    SparseSetup S(N, corr, C, T, 0.005f, weight_sparse);
    for (int k = 0; k < N; k++) { S.pRot[k] = make_float3(p[6 * k], p[6 * k + 1], p[6 * k + 2]); S.pTrans[k] = make_float3(p[6 * k + 3], p[6 * k + 4], p[6 * k + 5]); }
    for (int c = 0; c < C; c++) S.Jp[c] = applyJDevice((unsigned)c, S.in, S.st, S.prm);
    for (int k = 0; k < N; k++) {
        float3 accR = make_float3(0, 0, 0), accT = accR;
        if (k > 0)
            for (unsigned th = 0; th < THREADS_PER_BLOCK_JT; th++) {          // the block's 128 threads, one after the other
                float3 r, t;
                applyJTDevice((unsigned)k, S.in, S.st, S.prm, r, t, th, 0);
                accR += r; accT += t;
            }
        out[6 * k] = accR.x; out[6 * k + 1] = accR.y; out[6 * k + 2] = accR.z; out[6 * k + 3] = accT.x; out[6 * k + 4] = accT.y; out[6 * k + 5] = accT.z;
    }
}

// ---- dense term ----------------------------------------------------------------------------------------------
// JtJ [6N x 6N] and Jtr [6N] (per-frame order trans, rot) of the point-to-plane term for the ordered (target i, source j)
// pairs, BEFORE FlipJtJ; counts[p] = accepted pixels of pair p.
REF_API void ref_dense_system(int N, int Wd, int Hd, const float *intr, const float *campos, const float *normals, const float *T, const float *Tinv,
                              const int *pairs, int P, float dist_thresh, float normal_thresh, float depth_min, float depth_max, float robust_delta, float weight_dense,
                              float *JtJ, float *Jtr, int *counts)
{
    const unsigned npix = (unsigned)Wd * Hd, dim = 6u * N;
    memset(JtJ, 0, sizeof(float) * dim * dim); memset(Jtr, 0, sizeof(float) * dim);
    const float4 K = make_float4(intr[0], intr[1], intr[2], intr[3]);
    for (int pi = 0; pi < P; pi++) {
        const unsigned i = pairs[2 * pi], j = pairs[2 * pi + 1];
        // GLUE, SolverBundling.cu:143-148
        const float4x4 transform_i = load4(T + 16 * i), transform_j = load4(T + 16 * j), invTransform_i = load4(Tinv + 16 * i), invTransform_j = load4(Tinv + 16 * j);
        const float4x4 transform = invTransform_i * transform_j;
        const float4 *cam_i = reinterpret_cast<const float4 *>(campos) + (size_t)i * npix, *nrm_i = reinterpret_cast<const float4 *>(normals) + (size_t)i * npix;
        const float4 *cam_j = reinterpret_cast<const float4 *>(campos) + (size_t)j * npix, *nrm_j = reinterpret_cast<const float4 *>(normals) + (size_t)j * npix;
        int cnt = 0;
        for (unsigned srcIdx = 0; srcIdx < npix; srcIdx++) {
            matNxM<1, 6> row_i, row_j; row_i.setZero(); row_j.setZero();
            float3 camPosSrc, camPosSrcToTgt, camPosTgt, normalTgt; float2 tgtScreenPos;
            const bool found = findDenseCorr(srcIdx, (unsigned)Wd, (unsigned)Hd, dist_thresh, normal_thresh, transform, K, cam_i, nrm_i, cam_j, nrm_j,
                                             depth_min, depth_max, camPosSrc, camPosSrcToTgt, tgtScreenPos, camPosTgt, normalTgt);
            float res = 0.0f, weight = 0.0f;
            if (found) {
                // GLUE, SolverBundling.cu:176-187
                const float3 diff = camPosTgt - camPosSrcToTgt;
                res = dot(diff, normalTgt);
                float3 rho;
                huberLoss(res * res, robust_delta, rho);
                weight = weight_dense * rho.y;
                if (i > 0) computeJacobianBlockRow_i(row_i, transform_i, invTransform_j, camPosSrc, normalTgt);
                if (j > 0) computeJacobianBlockRow_j(row_j, invTransform_i, transform_j, camPosSrc, normalTgt);
                cnt++;
            }
            addToLocalSystemBrute(found, JtJ, Jtr, dim, row_i, row_j, i, j, res, weight, 0);
        }
        counts[pi] = cnt;
    }
}
