// oracle/ref_ransac_pre.h -- what /root/reference/src/cuda/cuda_ransac.cu needs before it starts (see ref_ransac_wrap.h and
// oracle/Makefile): its own header is skipped (Eigen, cuSOLVER, cuRAND), the matrix types come from the reference's
// cuda_SimpleMatrixUtil.h, the launches run on the sequential emulator of ref_shim/cuda_runtime.h.  Test infrastructure only.
#define __CUDA_RANSAC_H__
#define _CUTIL_INLINE_H_
#define _CUTIL_H_
#include <vector>
#include <math.h>
#include "cutil_math.h"
#include "cuda_SimpleMatrixUtil.h"
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }     // IEEE, never contracted
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __frsqrt_rn(float a)                  // correctly rounded, like the intrinsic: double result, settled exactly at the neighbouring float midpoints
{
    float r = (float)(1.0 / sqrt((double)a));
    unsigned rb; memcpy(&rb, &r, 4);
    if (!(r > 0.0f) || (rb & 0x7F800000u) == 0x7F800000u || (rb & 0x7F800000u) == 0u) return r;
    unsigned ub = rb + 1u, db = rb - 1u; float up, dn; memcpy(&up, &ub, 4); memcpy(&dn, &db, 4);
    const double mh = 0.5 * ((double)r + (double)up), ml = 0.5 * ((double)r + (double)dn);
    if (fma(mh * mh, (double)a, -1.0) < 0.0) return up;
    if (fma(ml * ml, (double)a, -1.0) > 0.0) return dn;
    return r;
}
// ransacGPU's (empty) signature mentions Eigen::Matrix4f
namespace Eigen { struct Matrix4f {}; }
// cuRAND -> oracle/xorwow.h (the published XORWOW algorithm restated; see that header for what pins it)
#include "xorwow.h"
typedef orc_xorwow_state curandState;
static inline void curand_init(unsigned long long seed, unsigned long long subsequence, unsigned long long offset, curandState *s) { orc_curand_init(seed, subsequence, offset, s); }
static inline float curand_uniform(curandState *s) { return orc_curand_uniform(s); }
// streams and atomicMax for the emulator: one thread at a time, every launch has completed when it returns
static inline cudaError_t cudaStreamCreate(cudaStream_t *s) { *s = 0; return 0; }
static inline cudaError_t cudaStreamSynchronize(cudaStream_t) { return 0; }
static inline cudaError_t cudaStreamDestroy(cudaStream_t) { return 0; }
static inline int atomicMax(int *p, int v) { const int o = *p; if (v > o) *p = v; return o; }
