/* oracle/ref_shim/curand.h -- intentionally empty stand-in (see cuda_runtime.h in this directory) */
