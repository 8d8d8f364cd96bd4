/* oracle/ref_shim/device_launch_parameters.h -- intentionally empty stand-in (see cuda_runtime.h in this directory) */
