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
#include <vector>
/* ---- execution order (round 6) -----------------------------------------------------------------------------------
 * The reference sums with float atomicAdd everywhere (SolverBundlingDenseUtil.h:217-285, SolverBundling.cu:575-818): on a GPU
 * the order in which the threads of a launch reach their atomics is arbitrary, and its own results move from run to run.
 * BTBA_REF_ORDER (environment, read at every launch; or ref_set_order() of the solver wrapper) chooses the order in which
 * the emulator walks the (block, thread) cells of EVERY launch:
 *     forward (default)   blocks then threads ascending -- the one order rounds 1-5 used
 *     reverse             the same walk backwards
 *     shuffle:<seed>      a seeded Fisher-Yates permutation of all cells, a fresh one per launch (seed advanced by a launch counter
 *                         that restarts at every ref_solve*, so a run is reproducible)
 * Any of these is a legal execution: the solve path's kernels meet only through atomics (see above).  Because the pair list
 * (FindImageImageCorr_Kernel) and the frame -> correspondence rows (BuildVariablesToCorrespondencesTableDevice) are handed out by
 * atomicAdd as well, their order moves too -- exactly as it may on the GPU. */
static int btba_order_mode = -1;            /* -1: not set by ref_set_order -> read BTBA_REF_ORDER; 0 forward, 1 reverse, 2 shuffle */
static unsigned long long btba_order_seed = 0, btba_launch_counter = 0;
static void (*btba_launch_hook)(const char *name) = 0;      /* called before every launch (the solver wrapper records the iterates with it) */
static inline unsigned long long btba_splitmix(unsigned long long &s) { unsigned long long z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static inline void btba_current_order(int &mode, unsigned long long &seed)
{
    mode = btba_order_mode; seed = btba_order_seed;
    if (mode < 0) {
        const char *e = getenv("BTBA_REF_ORDER");
        mode = 0;
        if (e && !strcmp(e, "reverse")) mode = 1;
        else if (e && !strncmp(e, "shuffle", 7)) { mode = 2; seed = e[7] == ':' ? strtoull(e + 8, 0, 10) : 1; }
    }
}
template <class F>
static void btba_emulate(const char *name, F body, dim3 grid, dim3 block, size_t = 0, void * = 0)
{
    if (strstr(name, "BuildDenseSystem_Kernel") || strstr(name, "FindDenseCorrespondences_Kernel")) { grid = dim3(grid.x, grid.y * block.x, 1); block = dim3(1, 1, 1); }
    gridDim = grid; blockDim = block;
    if (btba_launch_hook) btba_launch_hook(name);
    int mode; unsigned long long seed;
    btba_current_order(mode, seed);
    const unsigned long long launch = btba_launch_counter++;
    const size_t nthr = (size_t)block.x * block.y * block.z, nblk = (size_t)grid.x * grid.y * grid.z, total = nthr * nblk;
    /* cell c (forward order): block = c / nthr walked as (bz, bx, by), thread = c % nthr as (tz, ty, tx) -- the nesting rounds 1-5 had */
    auto run_cell = [&](size_t c) {
        size_t b = c / nthr, t = c % nthr;
        blockIdx.y = (unsigned)(b % grid.y); b /= grid.y; blockIdx.x = (unsigned)(b % grid.x); blockIdx.z = (unsigned)(b / grid.x);
        threadIdx.x = (unsigned)(t % block.x); t /= block.x; threadIdx.y = (unsigned)(t % block.y); threadIdx.z = (unsigned)(t / block.y);
        body();
    };
    if (mode == 0) { for (size_t c = 0; c < total; c++) run_cell(c); }
    else if (mode == 1) { for (size_t c = total; c-- > 0;) run_cell(c); }
    else {
        std::vector<unsigned> perm(total);
        for (size_t c = 0; c < total; c++) perm[c] = (unsigned)c;
        unsigned long long s = seed * 0xD1342543DE82EF95ull + launch * 0x2545F4914F6CDD1Dull + 1;
        for (size_t c = total; c > 1; c--) { size_t j = (size_t)(btba_splitmix(s) % c); unsigned tmp = perm[c - 1]; perm[c - 1] = perm[j]; perm[j] = tmp; }
        for (size_t c = 0; c < total; c++) run_cell(perm[c]);
    }
}
#define BTBA_LAUNCH(name, call, ...) btba_emulate(name, [&]() { call; }, __VA_ARGS__)

/* ---- a model of the reference's build flags (round 6; -DBTBA_REF_FASTMATH, the `_fm` libraries of oracle/Makefile) ------------
 * The reference is compiled with -use_fast_math (CMakeLists.txt:7) = --ftz=true --prec-div=false --prec-sqrt=false --fmad=true plus the
 * sin/cos intrinsics: its arithmetic is NOT IEEE.  The `_fm` build models what those flags license, with the host compiler's own
 * means where it has them and a seeded perturbation where it has not:
 *   fmad          -ffp-contract=fast -mfma                      (the compiler contracts a*b+c as nvcc does by default)
 *   prec-div=0,   -mrecip=all under -funsafe-math-optimizations -fno-associative-math -ffinite-math-only -fno-trapping-math:
 *   prec-sqrt=0   x / y = x * rcp(y), sqrt via rsqrt, each with one Newton step from the 12-bit estimate (~1-2 ulp) -- no reassociation
 *   ftz           MXCSR FTZ | DAZ for the duration of a ref_solve* call (ref_solver_wrap.h)
 *   sin / cos     the float result rounded from double, then moved by up to +-2 ulp chosen by a hash of (argument bits, BTBA_REF_FM_SEED):
 *                 __sinf / __cosf are less accurate than that (2^-21.4 absolute), so this UNDERSTATES the licence
 *   asin / acos   +-1 ulp likewise (software sequences built on the approximate division)
 * Different seeds are different, equally legal, libdevice builds. */
#ifdef BTBA_REF_FASTMATH
#include <cmath>
#include <iostream>
#include <algorithm>
#include <vector>
static unsigned long long btba_fm_seed = ~0ull;
static inline float btba_fm_perturb(float r, float arg, int max_ulp)
{
    if (btba_fm_seed == ~0ull) { const char *e = getenv("BTBA_REF_FM_SEED"); btba_fm_seed = e ? strtoull(e, 0, 10) : 1; }
    if (!(r == r) || r == 0.0f || fabsf(r) > 3.0e38f) return r;
    unsigned a; memcpy(&a, &arg, 4);
    unsigned long long s = btba_fm_seed * 0x9E3779B97F4A7C15ull + a;
    const int k = (int)(btba_splitmix(s) % (unsigned)(2 * max_ulp + 1)) - max_ulp;
    int bits; memcpy(&bits, &r, 4); bits += (bits < 0) ? -k : k; memcpy(&r, &bits, 4);      /* k ulp away from zero (k > 0) or towards it: the sign does not matter for a bound */
    return r;
}
static inline float  btba_fm_sin(float x)   { return btba_fm_perturb((float)::sin((double)x), x, 2); }
static inline float  btba_fm_cos(float x)   { return btba_fm_perturb((float)::cos((double)x), x, 2); }
static inline float  btba_fm_asin(float x)  { return btba_fm_perturb((float)::asin((double)x), x, 1); }
static inline float  btba_fm_acos(float x)  { return btba_fm_perturb((float)::acos((double)x), x, 1); }
static inline double btba_fm_sin(double x)  { return ::sin(x); }
static inline double btba_fm_cos(double x)  { return ::cos(x); }
static inline double btba_fm_asin(double x) { return ::asin(x); }
static inline double btba_fm_acos(double x) { return ::acos(x); }
#define sin(x)   btba_fm_sin(x)
#define cos(x)   btba_fm_cos(x)
#define asin(x)  btba_fm_asin(x)
#define acos(x)  btba_fm_acos(x)
#define sinf(x)  btba_fm_sin((float)(x))
#define cosf(x)  btba_fm_cos((float)(x))
#define asinf(x) btba_fm_asin((float)(x))
#define acosf(x) btba_fm_acos((float)(x))
#endif
#endif
