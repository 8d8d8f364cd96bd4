// oracle/ref_ransac_wrap.h -- C ABI around the reference's own RANSAC code (cuda_ransac.cu): procrustesKernel / evalPoseKernel
// (:978-1102) as functions, and ransacMultiPairGPU (:1228-1323) with its three kernels end to end.  Appended, by oracle/Makefile's
// `ref` target, to /root/reference/src/cuda/cuda_ransac.cu as it is streamed from the reference checkout into the compiler at
// build time (launches rewritten for the sequential emulator, cuRAND mapped onto oracle/xorwow.h by ref_ransac_pre.h); nothing is
// copied into this repository.  Test infrastructure only.
// Emulation notes: threads run one after the other in launch order, so (1) ransacEvalModelKernel's atomicAdd counts are exact, and
// (2) findBestTrial -- atomicMax, a block-local __syncthreads, then "whoever equals the maximum writes its id", a race on the GPU
// -- resolves to the LAST trial that reaches the running maximum, i.e. the highest trial id among the best.  A GPU run may pick any
// of the tied trials; the product and the oracle pick the lowest id.  Tests compare modulo that tie rule.
extern "C" __attribute__((visibility("default")))
int ref_procrustes(const float *src, const float *dst, int n_pts, float *pose_rowmajor)
{
    float4x4 P;
    const bool ok = procrustesKernel(reinterpret_cast<const float4 *>(src), reinterpret_cast<const float4 *>(dst), n_pts, P);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) pose_rowmajor[4 * r + c] = P(r, c);
    return ok ? 1 : 0;
}
extern "C" __attribute__((visibility("default")))
int ref_eval_pose(const float *ptsA, const float *ptsB, int n_pts, const float *pose_rowmajor, float dist_thres, int *inlier_ids)
{
    float4x4 P;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) P(r, c) = pose_rowmajor[4 * r + c];
    return evalPoseKernel(reinterpret_cast<const float4 *>(ptsA), reinterpret_cast<const float4 *>(ptsB), n_pts, P, dist_thres, inlier_ids);
}

// ransacMultiPairGPU on pairs laid out back to back (pair p owns n_pts[p] float4 points from offset sum(n_pts[:p])):
// inlier_ids_out in the same layout, n_inliers_out[p] entries valid per pair.
extern "C" __attribute__((visibility("default")))
int ref_ransac_multi_pair(const float *ptsA_all, const float *ptsB_all, const int *n_pts, int n_pairs, int n_trials, float dist_thres,
                          int *inlier_ids_out, int *n_inliers_out)
{
    std::vector<float4 *> A(n_pairs), B(n_pairs);
    std::vector<int> n(n_pts, n_pts + n_pairs);
    size_t o = 0;
    for (int p = 0; p < n_pairs; p++) {
        A[p] = const_cast<float4 *>(reinterpret_cast<const float4 *>(ptsA_all) + o);
        B[p] = const_cast<float4 *>(reinterpret_cast<const float4 *>(ptsB_all) + o);
        o += (size_t)n_pts[p];
    }
    std::vector<std::vector<int>> ids;
    ransacMultiPairGPU(A, B, n, n_trials, dist_thres, ids);
    o = 0;
    for (int p = 0; p < n_pairs; p++) {
        n_inliers_out[p] = (int)ids[p].size();
        for (size_t k = 0; k < ids[p].size(); k++) inlier_ids_out[o + k] = ids[p][k];
        o += (size_t)n_pts[p];
    }
    return 0;
}
