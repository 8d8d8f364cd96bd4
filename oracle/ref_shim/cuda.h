/* oracle/ref_shim/cuda.h -- intentionally empty stand-in (see cuda_runtime.h in this directory) */
