/* oracle/ref_shim/device_functions.h -- intentionally empty stand-in (see cuda_runtime.h in this directory) */
