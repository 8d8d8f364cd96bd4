/* oracle/ref_shim/math_constants.h -- intentionally empty stand-in (see cuda_runtime.h in this directory) */
#ifndef BTBA_REF_SHIM_MATH_CONSTANTS_H
#define BTBA_REF_SHIM_MATH_CONSTANTS_H
#define CUDART_PI_F 3.141592654f
#define CUDART_INF_F (__int_as_float(0x7f800000))
#define CUDART_NAN_F (__int_as_float(0x7fffffff))
#endif
