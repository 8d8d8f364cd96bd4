/* oracle/ref_shim/cuda_runtime_api.h -- intentionally empty stand-in (see cuda_runtime.h in this directory) */
