// oracle/ref_solver_wrap.h -- C ABI around the reference's WHOLE solver, run on the CPU.
// Appended by oracle/Makefile (target ref) to /root/reference/src/cuda/Solver/SolverBundling.cu after its kernel launches
// have been rewritten for the sequential emulator in ref_shim/cuda_runtime.h (nothing of the reference is copied into the
// repository; the file is streamed into the compiler).  What runs is the reference's own code: solveBundlingStub and every
// kernel and __device__ function under it (BuildDenseSystem, FlipJtJ, PCGInit, PCGStep_Kernel0..3, the dense mat-vec,
// computeLieUpdate, convertLiePosesToMatricesCU, the frame->correspondence table).  GLUE restated here, because it lives
// in host files that need Eigen / yaml-cpp / mLib: the set-up of SolverInput / SolverState / SolverParameters
// (CUDASolverBundling.cpp:22-141, 190-241, with SBA.cpp:27-32, 81-117) and the matrix <-> se(3) conversion at both ends,
// done with the reference's matrixToPose / poseToMatrix.  Test infrastructure only.
#include <vector>

static float4x4 rs_load4(const float *m) { float4x4 M; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M(r, c) = m[4 * r + c]; return M; }

// addr_rank (may be NULL): a permutation of 0 .. N-1 giving the ORDER of the frames' d_num_valid_points addresses -- frame k's
// pointer is &nvalid[addr_rank[k]], so frame a's address is above frame b's iff addr_rank[a] > addr_rank[b].
// FindImageImageCorr_Kernel (SolverBundling.cu:17-47) keeps the ordered pair (i, j) = (target, source) iff address_i > address_j:
// NULL = descending addresses in frame order (target = lower index, BTBA_PAIRS_TARGET_LOWER); addr_rank[k] = k is what one
// cudaMalloc per frame in a fresh CUDACache typically yields (ascending: target = higher index, every dense cross block written
// above the diagonal and erased by FlipJtJ_Kernel, :49-59 -- BTBA_PAIRS_TARGET_HIGHER); any other permutation gives the
// corresponding explicit orientation (BTBA_PAIRS_TARGET_MORE_VALID, BTBA_PAIRS_EXPLICIT).
// ---- round 6: execution order, build-flag model, per-iterate record ---------------------------------------------------------
// ref_set_order(mode, seed): 0 forward, 1 reverse, 2 seeded shuffle of every launch's (block, thread) cells, -1 = follow the
// environment variable BTBA_REF_ORDER (ref_shim/cuda_runtime.h).  ref_build_flags(): 1 for the `_fm` library (-DBTBA_REF_FASTMATH).
extern "C" __attribute__((visibility("default"))) void ref_set_order(int mode, unsigned long long seed) { btba_order_mode = mode; btba_order_seed = seed; }
extern "C" __attribute__((visibility("default"))) int ref_build_flags(void)
{
#ifdef BTBA_REF_FASTMATH
    return 1;
#else
    return 0;
#endif
}
#ifdef BTBA_REF_FASTMATH
extern "C" __attribute__((visibility("default"))) void ref_set_fastmath_seed(unsigned long long seed) { btba_fm_seed = seed; }
#include <xmmintrin.h>
#endif
// The iterates: solveBundlingStub launches convertLiePosesToMatricesCU_Kernel at the top of every Gauss-Newton iteration
// (SolverBundling.cu:953) with the CURRENT (d_xRot, d_xTrans); the emulator's launch hook copies them, so x_iter[n] = the unknowns
// after iteration n (the copy taken at the top of iteration n + 1; the last one after the stub returns).
static const float3 *rs_trace_rot = nullptr, *rs_trace_trans = nullptr;
static float *rs_trace_out = nullptr;
static int rs_trace_n = 0, rs_trace_seen = 0, rs_trace_max = 0;
static void rs_trace_hook(const char *name)
{
    if (!rs_trace_out || !strstr(name, "convertLiePosesToMatricesCU_Kernel")) return;
    if (rs_trace_seen > 0 && rs_trace_seen <= rs_trace_max) {
        float *o = rs_trace_out + (size_t)(rs_trace_seen - 1) * rs_trace_n * 6;
        for (int k = 0; k < rs_trace_n; k++) { o[6 * k] = rs_trace_rot[k].x; o[6 * k + 1] = rs_trace_rot[k].y; o[6 * k + 2] = rs_trace_rot[k].z;
                                               o[6 * k + 3] = rs_trace_trans[k].x; o[6 * k + 4] = rs_trace_trans[k].y; o[6 * k + 5] = rs_trace_trans[k].z; }
    }
    rs_trace_seen++;
}

extern "C" __attribute__((visibility("default")))
int ref_solve4(int N, int Wd, int Hd, const float *intr, const float *campos, const float *normals, const float *corr_in, int C,
               float *poses_io /* [N][16] row-major, camera -> model */, int n_gn, int n_pcg, float w_sparse, float w_dense, float robust_delta,
               float dist_thresh, float normal_thresh, float depth_min, float depth_max, float *x_out /* [N][6] rot, trans; may be NULL */,
               const int *addr_rank, const float *w_sparse_it /* [n_gn] or NULL: w_sparse in every iteration */, const float *w_dense_it /* likewise */,
               float *T_iter /* [n_gn][N][16]: poseToMatrix of the unknowns after every iteration; may be NULL */)
{
    btba_launch_counter = 0;                                 // a run is reproducible whatever ran before it in this process
#ifdef BTBA_REF_FASTMATH
    const unsigned csr_saved = _mm_getcsr();
    _mm_setcsr(csr_saved | 0x8040u);                         // --ftz=true: flush-to-zero + denormals-are-zero
#endif
    std::vector<float> x_iter(T_iter ? (size_t)n_gn * N * 6 : 0);
    const size_t npix = (size_t)Wd * Hd;
    const unsigned maxCorrPerImage = C > 0 ? (unsigned)C : 1u, maxPairs = (unsigned)(N * (N - 1) / 2 > 0 ? N * (N - 1) / 2 : 1);
    // frames: CUDACachedFrame[] with the float4 camPos / normal maps; the orientation of a dense pair is decided by comparing the
    // frames' d_num_valid_points POINTERS (FindImageImageCorr_Kernel, SolverBundling.cu:25-33): addr_rank chooses their order
    std::vector<CUDACachedFrame> frames(N);
    std::vector<int> nvalid(N, 0);
    std::vector<float> depth(npix * N);
    for (int k = 0; k < N; k++) {
        memset(&frames[k], 0, sizeof(CUDACachedFrame));
        frames[k].d_cameraposDownsampled = const_cast<float4 *>(reinterpret_cast<const float4 *>(campos) + k * npix);
        frames[k].d_normalsDownsampled = const_cast<float4 *>(reinterpret_cast<const float4 *>(normals) + k * npix);
        for (size_t q = 0; q < npix; q++) depth[k * npix + q] = campos[4 * (k * npix + q) + 2];
        frames[k].d_depthDownsampled = depth.data() + k * npix;
        frames[k].d_num_valid_points = &nvalid[addr_rank ? addr_rank[k] : N - 1 - k];
    }
    std::vector<EntryJ> corr(C > 0 ? C : 1);
    if (C > 0) memcpy(corr.data(), corr_in, sizeof(EntryJ) * (size_t)C);
    std::vector<int> valid(N, 1), v2c((size_t)N * maxCorrPerImage, 0), nrows(N, 0);
    std::vector<float3> xRot(N), xTrans(N);
    for (int k = 0; k < N; k++) matrixToPose(rs_load4(poses_io + 16 * k), xRot[k], xTrans[k]);          // convertMatricesToPosesCU, SBA.cpp:106

    SolverState st; memset(&st, 0, sizeof st);
    auto f3 = [&](size_t n) { return (float3 *)calloc(n ? n : 1, sizeof(float3)); };
    st.d_deltaRot = f3(N); st.d_deltaTrans = f3(N); st.d_rRot = f3(N); st.d_rTrans = f3(N); st.d_zRot = f3(N); st.d_zTrans = f3(N);
    st.d_pRot = f3(N); st.d_pTrans = f3(N); st.d_Jp = f3(C); st.d_Ap_XRot = f3(N); st.d_Ap_XTrans = f3(N);
    st.d_precondionerRot = f3(N); st.d_precondionerTrans = f3(N);
    st.d_scanAlpha = (float *)calloc(2, 4); st.d_rDotzOld = (float *)calloc(N, 4); st.d_sumResidual = (float *)calloc(1, 4);
    st.d_countHighResidual = (int *)calloc(1, 4);
    st.d_denseJtJ = (float *)calloc((size_t)36 * N * N, 4); st.d_denseJtr = (float *)calloc((size_t)6 * N, 4);
    st.d_denseCorrCounts = (float *)calloc(maxPairs, 4);
    st.d_xTransforms = (float4x4 *)calloc(N, sizeof(float4x4)); st.d_xTransformInverses = (float4x4 *)calloc(N, sizeof(float4x4));
    st.d_denseOverlappingImages = (uint2 *)calloc(maxPairs, sizeof(uint2)); st.d_numDenseOverlappingImages = (int *)calloc(1, 4);
    st.d_corrCount = (int *)calloc(1, 4); st.d_corrCountColor = (int *)calloc(1, 4); st.d_sumResidualColor = (float *)calloc(1, 4);
    st.d_xRot = xRot.data(); st.d_xTrans = xTrans.data();

    std::vector<float> wS(n_gn, w_sparse), wD(n_gn, w_dense), wC(n_gn, 0.0f);                              // SBA.cpp:27-32
    for (int it = 0; it < n_gn; it++) { if (w_sparse_it) wS[it] = w_sparse_it[it]; if (w_dense_it) wD[it] = w_dense_it[it]; }      // input.weightsSparse / weightsDenseDepth, read per iteration at SolverBundling.cu:948-949
    SolverParameters prm; memset(&prm, 0, sizeof prm);
    prm.denseDistThresh = dist_thresh; prm.denseNormalThresh = normal_thresh; prm.denseColorThresh = 0.1f; prm.denseColorGradientMin = 0.005f;
    prm.denseDepthMin = depth_min; prm.denseDepthMax = depth_max; prm.denseOverlapCheckSubsampleFactor = 1;  // CUDASolverBundling.cpp:93-99
    prm.nNonLinearIterations = n_gn; prm.nLinIterations = n_pcg; prm.robust_delta = robust_delta;
    prm.highResidualThresh = INFINITY;
    prm.weightSparse = wS[0]; prm.weightDenseDepth = wD[0]; prm.weightDenseColor = wC[0];
    prm.useDense = (prm.weightDenseDepth > 0 || prm.weightDenseColor > 0); prm.useDenseDepthAllPairwise = true;   // SBA.cpp:90
    SolverInput in; memset(&in, 0, sizeof in);
    in.d_correspondences = corr.data(); in.d_variablesToCorrespondences = v2c.data(); in.d_numEntriesPerRow = nrows.data();
    in.numberOfImages = N; in.numberOfCorrespondences = C; in.maxNumberOfImages = N; in.maxCorrPerImage = maxCorrPerImage; in.maxNumDenseImPairs = maxPairs;
    in.weightsSparse = wS.data(); in.weightsDenseDepth = wD.data(); in.weightsDenseColor = wC.data();
    in.d_validImages = valid.data(); in.d_cacheFrames = frames.data(); in.denseDepthWidth = Wd; in.denseDepthHeight = Hd;
    in.intrinsics = make_float4(intr[0], intr[1], intr[2], intr[3]);
    in.colorFocalLength = make_float2(intr[0], intr[1]);

    if (C > 0) buildVariablesToCorrespondencesTableCUDA(corr.data(), (unsigned)C, maxCorrPerImage, v2c.data(), nrows.data(), nullptr);
    SolverStateAnalysis analysis; memset(&analysis, 0, sizeof analysis);
    rs_trace_rot = xRot.data(); rs_trace_trans = xTrans.data(); rs_trace_out = T_iter ? x_iter.data() : nullptr; rs_trace_n = N; rs_trace_seen = 0; rs_trace_max = n_gn;
    btba_launch_hook = rs_trace_hook;
    solveBundlingStub(in, st, prm, analysis, nullptr, nullptr);
    btba_launch_hook = nullptr;
    if (T_iter) {
        rs_trace_seen = n_gn; rs_trace_hook("convertLiePosesToMatricesCU_Kernel");                           // the last iterate
        for (int it = 0; it < n_gn; it++) for (int k = 0; k < N; k++) {
            const float *o = x_iter.data() + ((size_t)it * N + k) * 6;
            const float4x4 M = poseToMatrix(make_float3(o[0], o[1], o[2]), make_float3(o[3], o[4], o[5]));
            for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T_iter[((size_t)it * N + k) * 16 + 4 * r + c] = M(r, c);
        }
    }
    rs_trace_out = nullptr;

    for (int k = 0; k < N; k++) {                                                                           // convertPosesToMatricesCU, SBA.cpp:115
        const float4x4 M = poseToMatrix(xRot[k], xTrans[k]);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) poses_io[16 * k + 4 * r + c] = M(r, c);
        if (x_out) { x_out[6 * k] = xRot[k].x; x_out[6 * k + 1] = xRot[k].y; x_out[6 * k + 2] = xRot[k].z; x_out[6 * k + 3] = xTrans[k].x; x_out[6 * k + 4] = xTrans[k].y; x_out[6 * k + 5] = xTrans[k].z; }
    }
    void *to_free[] = { st.d_deltaRot, st.d_deltaTrans, st.d_rRot, st.d_rTrans, st.d_zRot, st.d_zTrans, st.d_pRot, st.d_pTrans, st.d_Jp, st.d_Ap_XRot, st.d_Ap_XTrans,
                        st.d_precondionerRot, st.d_precondionerTrans, st.d_scanAlpha, st.d_rDotzOld, st.d_sumResidual, st.d_countHighResidual, st.d_denseJtJ, st.d_denseJtr,
                        st.d_denseCorrCounts, st.d_xTransforms, st.d_xTransformInverses, st.d_denseOverlappingImages, st.d_numDenseOverlappingImages, st.d_corrCount,
                        st.d_corrCountColor, st.d_sumResidualColor };
    for (void *q : to_free) free(q);
#ifdef BTBA_REF_FASTMATH
    _mm_setcsr(csr_saved);
#endif
    return 0;
}

extern "C" __attribute__((visibility("default")))
int ref_solve3(int N, int Wd, int Hd, const float *intr, const float *campos, const float *normals, const float *corr_in, int C,
               float *poses_io, int n_gn, int n_pcg, float w_sparse, float w_dense, float robust_delta,
               float dist_thresh, float normal_thresh, float depth_min, float depth_max, float *x_out, const int *addr_rank,
               const float *w_sparse_it, const float *w_dense_it)
{
    return ref_solve4(N, Wd, Hd, intr, campos, normals, corr_in, C, poses_io, n_gn, n_pcg, w_sparse, w_dense, robust_delta, dist_thresh, normal_thresh,
                      depth_min, depth_max, x_out, addr_rank, w_sparse_it, w_dense_it, nullptr);
}

extern "C" __attribute__((visibility("default")))
int ref_solve2(int N, int Wd, int Hd, const float *intr, const float *campos, const float *normals, const float *corr_in, int C,
               float *poses_io, int n_gn, int n_pcg, float w_sparse, float w_dense, float robust_delta,
               float dist_thresh, float normal_thresh, float depth_min, float depth_max, float *x_out, const int *addr_rank)
{
    return ref_solve3(N, Wd, Hd, intr, campos, normals, corr_in, C, poses_io, n_gn, n_pcg, w_sparse, w_dense, robust_delta, dist_thresh, normal_thresh,
                      depth_min, depth_max, x_out, addr_rank, nullptr, nullptr);
}

extern "C" __attribute__((visibility("default")))
int ref_solve(int N, int Wd, int Hd, const float *intr, const float *campos, const float *normals, const float *corr_in, int C,
              float *poses_io, int n_gn, int n_pcg, float w_sparse, float w_dense, float robust_delta,
              float dist_thresh, float normal_thresh, float depth_min, float depth_max, float *x_out)
{
    return ref_solve2(N, Wd, Hd, intr, campos, normals, corr_in, C, poses_io, n_gn, n_pcg, w_sparse, w_dense, robust_delta, dist_thresh, normal_thresh,
                      depth_min, depth_max, x_out, nullptr);
}
