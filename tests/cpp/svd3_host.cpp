// Host build of bundletrack_amd/csrc/btba_svd3.hpp for the CPU tests: the product's restatement of the reference's 3x3 SVD and
// procrustesKernel, compiled with g++ (-ffp-contract=off), held bit for bit against the reference's own functions.
#include "../../bundletrack_amd/csrc/btba_svd3.hpp"

struct P4 { float x, y, z, w; };

extern "C" __attribute__((visibility("default"))) void svd3_host(const float *A, float *U, float *s, float *V)
{
    float a[9], u[9], v[9], sg[3];
    for (int k = 0; k < 9; k++) a[k] = A[k];
    btba::svd3::svd(a, u, sg, v);
    for (int k = 0; k < 9; k++) { U[k] = u[k]; V[k] = v[k]; }
    for (int k = 0; k < 3; k++) s[k] = sg[k];
}

extern "C" __attribute__((visibility("default"))) int procrustes_reference_host(const float *src, const float *dst, int n, float *pose16)
{
    float P[12];
    const bool ok = btba::svd3::procrustes_reference(reinterpret_cast<const P4 *>(src), reinterpret_cast<const P4 *>(dst), n, P);
    for (int k = 0; k < 12; k++) pose16[k] = P[k];
    pose16[12] = pose16[13] = pose16[14] = 0.0f; pose16[15] = 1.0f;
    return ok ? 1 : 0;
}

// rsqrt_rn over an array (tests/test_oracle_ransac.py::test_rsqrt_is_correctly_rounded)
extern "C" __attribute__((visibility("default"))) void rsqrt_rn_host(const float *x, int n, float *out)
{
    for (int k = 0; k < n; k++) out[k] = btba::svd3::rsqrt_rn(x[k]);
}
