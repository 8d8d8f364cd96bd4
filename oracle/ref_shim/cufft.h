/* oracle/ref_shim/cufft.h -- intentionally empty stand-in (see cuda_runtime.h in this directory) */
