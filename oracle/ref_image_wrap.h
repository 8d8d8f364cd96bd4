// oracle/ref_image_wrap.h -- C ABI around the reference's image kernels (src/cuda/CUDAImageUtil.cu), run on the CPU through the
// sequential launch emulator (see ref_shim/cuda_runtime.h and ref_solver_wrap.h).  Appended by oracle/Makefile to the
// streamed, launch-rewritten CUDAImageUtil.cu; nothing of the reference is copied.  GLUE restated: the call sequences of
// CUDACache::storeFrame (CUDACache.cpp:76-88), Frame::processDepth and Frame::depthToCloudAndNormals (Frame.cpp:152-233).
// Test infrastructure only.
static float4x4 ri_load4(const float *m) { float4x4 M; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M(r, c) = m[4 * r + c]; return M; }
#define REF_IMG_API extern "C" __attribute__((visibility("default")))

// CUDACache::storeFrame for one frame: depth [H*W], normals float4 [H*W], full-resolution intrinsicsInv (4x4 row-major)
REF_IMG_API void ref_store_frame(int W, int H, int Wd, int Hd, const float *Kinv, const float *depth, const float *normals,
                                 float *campos_out /* float4 [Hd*Wd] */, float *normals_out, float *depth_out, int *n_valid)
{
    float4 *helper = (float4 *)calloc((size_t)W * H, sizeof(float4));
    CUDAImageUtil::convertDepthFloatToCameraSpaceFloat4(helper, depth, ri_load4(Kinv), W, H);
    CUDAImageUtil::resampleFloat4(reinterpret_cast<float4 *>(campos_out), Wd, Hd, helper, W, H);
    CUDAImageUtil::resampleFloat4(reinterpret_cast<float4 *>(normals_out), Wd, Hd, reinterpret_cast<const float4 *>(normals), W, H);
    CUDAImageUtil::resampleFloat(depth_out, Wd, Hd, depth, W, H);
    *n_valid = 0;
    CUDAImageUtil::countNumValidDepth(n_valid, depth_out, Hd, Wd);
    free(helper);
}
// Frame::processDepth: erode, then the depth "gauss" (mean-gated bilateral) filter twice
REF_IMG_API void ref_process_depth(int W, int H, const float *in, float *out, int erode_radius, float erode_diff, float erode_ratio, int bf_radius, float sigma_d, float sigma_r)
{
    float *a = (float *)calloc((size_t)W * H, 4), *b = (float *)calloc((size_t)W * H, 4);
    memcpy(a, in, sizeof(float) * (size_t)W * H);
    CUDAImageUtil::erodeDepthMap(b, a, erode_radius, W, H, erode_diff, erode_ratio);
    CUDAImageUtil::gaussFilterDepthMap(a, b, bf_radius, sigma_d, sigma_r, W, H);
    CUDAImageUtil::gaussFilterDepthMap(b, a, bf_radius, sigma_d, sigma_r, W, H);
    memcpy(out, b, sizeof(float) * (size_t)W * H);
    free(a); free(b);
}
REF_IMG_API void ref_erode(int W, int H, const float *in, float *out, int radius, float diff, float ratio)
{
    CUDAImageUtil::erodeDepthMap(out, const_cast<float *>(in), radius, W, H, diff, ratio);
}
REF_IMG_API void ref_gauss_filter(int W, int H, const float *in, float *out, int radius, float sigma_d, float sigma_r)
{
    CUDAImageUtil::gaussFilterDepthMap(out, in, radius, sigma_d, sigma_r, W, H);
}
// Frame::depthToCloudAndNormals
REF_IMG_API void ref_depth_to_normals(int W, int H, const float *Kinv, const float *depth, float *normals_out, float *xyz_out)
{
    float4 *xyz = (float4 *)calloc((size_t)W * H, sizeof(float4));
    CUDAImageUtil::convertDepthFloatToCameraSpaceFloat4(xyz, depth, ri_load4(Kinv), W, H);
    CUDAImageUtil::computeNormals(reinterpret_cast<float4 *>(normals_out), xyz, W, H);
    if (xyz_out) memcpy(xyz_out, xyz, sizeof(float4) * (size_t)W * H);
    free(xyz);
}
