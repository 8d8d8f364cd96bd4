// oracle/ref_ransac_wrap.h -- C ABI around the reference's own procrustesKernel / evalPoseKernel (cuda_ransac.cu:978-1102).
// Appended, by oracle/Makefile's `ref` target, to the first 1103 lines of /root/reference/src/cuda/cuda_ransac.cu -- the
// part that holds the __device__ functions (McAdams' 3x3 SVD, evalPoseKernel, procrustesKernel); the kernels and the host
// launcher below that line use <<< >>> and cuRAND and are not compiled.  The slice is streamed from the reference
// checkout into the compiler at build time; nothing is copied into this repository.  Test infrastructure only.
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
