// oracle/ref_ransac_pre.h -- what the slice of cuda_ransac.cu needs before it starts (see ref_ransac_wrap.h):
// its own header is skipped (Eigen, cuSOLVER, cuRAND), the matrix types come from the reference's cuda_SimpleMatrixUtil.h.
#define __CUDA_RANSAC_H__
#define _CUTIL_INLINE_H_
#define _CUTIL_H_
#include <vector>
#include "cutil_math.h"
#include "cuda_SimpleMatrixUtil.h"
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }     // IEEE, never contracted
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __frsqrt_rn(float a) { return (float)(1.0 / sqrt((double)a)); }         // correctly rounded, like the intrinsic
