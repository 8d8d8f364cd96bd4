/* oracle/ref_shim/core-util/timer.h -- empty stand-in: cuda_ransac.cu includes mLib's timer and never uses it in the compiled part */
