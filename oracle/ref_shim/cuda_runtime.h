/* oracle/ref_shim/cuda_runtime.h -- NOT the CUDA runtime.  The handful of CUDA built-ins that the reference's
 * header-only math (cutil_math.h, cuda_SimpleMatrixUtil.h, Solver/LieDerivUtil.h, ICPUtil.h, SolverBundlingUtil.h,
 * SolverBundlingEquationsLie.h) touches, so that those files -- read where they lie under /root/reference, never
 * copied -- compile with g++ and can be CALLED on the CPU to pin the oracle (oracle/Makefile, target ref).
 * Test infrastructure only. */
#ifndef BTBA_REF_SHIM_CUDA_RUNTIME_H
#define BTBA_REF_SHIM_CUDA_RUNTIME_H
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __constant__ static
#define __restrict__

typedef unsigned int uint;
typedef unsigned short ushort;
typedef unsigned char uchar;

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct uint4 { unsigned x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

static inline float2 make_float2(float x, float y) { float2 v = { x, y }; return v; }
static inline float3 make_float3(float x, float y, float z) { float3 v = { x, y, z }; return v; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 v = { x, y, z, w }; return v; }
static inline int2 make_int2(int x, int y) { int2 v = { x, y }; return v; }
static inline int3 make_int3(int x, int y, int z) { int3 v = { x, y, z }; return v; }
static inline int4 make_int4(int x, int y, int z, int w) { int4 v = { x, y, z, w }; return v; }
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 v = { x, y }; return v; }
static inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { uint3 v = { x, y, z }; return v; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 v = { x, y, z, w }; return v; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { uchar4 v = { x, y, z, w }; return v; }

static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __saturatef(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }
using std::max;
using std::min;

/* thread geometry: ONE thread runs at a time; btba_emulate() below walks a launch's blocks and threads sequentially */
static uint3 threadIdx = { 0, 0, 0 }, blockIdx = { 0, 0, 0 };
static dim3 blockDim(1, 1, 1), gridDim(1, 1, 1);
static inline void __syncthreads() {}
/* one lane at a time: the other lanes of the "warp" contribute nothing, so the reference's warpReduce(v) returns v and the
 * caller (oracle/ref_driver.cpp) adds up the lanes itself */
template <class T> static inline T __shfl_down_sync(unsigned, T, int, int = 32) { return T(0); }
template <class T, class U> static inline T atomicAdd(T *p, U v) { T o = *p; *p += (T)v; return o; }

/* "device" memory is host memory here; the runtime calls the headers' helper structs make become libc calls */
typedef int cudaError_t;
static const cudaError_t cudaSuccess = 0;
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
static inline const char *cudaGetErrorString(cudaError_t) { return "shim"; }
template <class T> static inline cudaError_t cudaMalloc(T **p, size_t n) { *p = (T *)calloc(1, n ? n : 1); return *p ? 0 : 2; }
static inline cudaError_t cudaFree(void *p) { free(p); return 0; }
static inline cudaError_t cudaMemset(void *p, int v, size_t n) { memset(p, v, n); return 0; }
static inline cudaError_t cudaMemcpy(void *d, const void *s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return 0; }
static inline cudaError_t cudaDeviceSynchronize() { return 0; }
static inline cudaError_t cudaGetLastError() { return 0; }
typedef void *cudaEvent_t;
typedef void *cudaStream_t;
static inline cudaError_t cudaEventCreate(cudaEvent_t *e) { *e = 0; return 0; }
static const unsigned cudaEventBlockingSync = 1;
static inline cudaError_t cudaEventCreateWithFlags(cudaEvent_t *e, unsigned) { *e = 0; return 0; }
static inline cudaError_t cudaEventDestroy(cudaEvent_t) { return 0; }
static inline cudaError_t cudaEventRecord(cudaEvent_t, cudaStream_t = 0) { return 0; }
static inline cudaError_t cudaEventSynchronize(cudaEvent_t) { return 0; }
static inline cudaError_t cudaEventElapsedTime(float *ms, cudaEvent_t, cudaEvent_t) { *ms = 0.f; return 0; }
#define cutilSafeCall(x) (x)
#define cutilCheckMsg(x)

/* ---- kernel launches ------------------------------------------------------------------------------------------
 * oracle/Makefile rewrites `kernel<<<grid, block>>>(args);` in the streamed .cu into BTBA_LAUNCH("kernel", (kernel(args)),
 * grid, block); the kernel (a plain function here, __global__ is empty) is then called once per (block, thread), in order.
 * That is a faithful execution for kernels whose threads only meet through atomics -- every kernel of the solve path once
 * WARP_SIZE is 1 in the .cu (each thread is lane 0 of its own warp, so the `lane == 0 -> atomicAdd(warpReduce(v))`
 * pattern adds every thread's value).  The two dense-sweep kernels also reduce through __shared__ memory between
 * __syncthreads(); they index pixels as threadIdx.x * gridDim.y + blockIdx.y, so they are run with one thread per block
 * and gridDim.y enlarged by the block size: the same set of pixels, and the block-wide reduction degenerates to the
 * thread's own value.  Sums are therefore formed sequentially in launch order (the GPU's order is arbitrary). */
#include <string.h>
template <class F>
static void btba_emulate(const char *name, F body, dim3 grid, dim3 block, size_t = 0, void * = 0)
{
    if (strstr(name, "BuildDenseSystem_Kernel") || strstr(name, "FindDenseCorrespondences_Kernel")) { grid = dim3(grid.x, grid.y * block.x, 1); block = dim3(1, 1, 1); }
    gridDim = grid; blockDim = block;
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned bx = 0; bx < grid.x; bx++) for (unsigned by = 0; by < grid.y; by++)
        for (unsigned tz = 0; tz < block.z; tz++) for (unsigned ty = 0; ty < block.y; ty++) for (unsigned tx = 0; tx < block.x; tx++) {
            blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz; threadIdx.x = tx; threadIdx.y = ty; threadIdx.z = tz;
            body();
        }
}
#define BTBA_LAUNCH(name, call, ...) btba_emulate(name, [&]() { call; }, __VA_ARGS__)
#endif
