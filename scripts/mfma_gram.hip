// mfma_gram.hip -- the measured answer to "should the dense sweep's Gram update run on the matrix cores?"  (round 4, verdict item 4)
//
// Per accepted pixel the dense sweep adds  w [a | res] [a | res]^T  (a = the 6-entry Jacobian row, res the residual: 21 + 6 distinct products,
// btba_kernels.hpp: dense_block_pinhole) to 28 per-lane accumulators: 7 multiplies for w a_r and 27 fused multiply-adds, ~35 vector
// instructions of a ~144-instruction pixel.  This microbenchmark isolates exactly that stage on synthetic rows and times two forms of it at the
// sweep's occupancy (256-thread workgroups, launch bound 6 waves per SIMD, grid = 6 workgroups per compute unit):
//
//   mode 0  VALU    the production code: (w a_r, a_r) register pairs, 27 v_fma_f32 + 1 count per pixel, 28 accumulators per lane
//   mode 1  MFMA    v_mfma_f32_16x16x4_f32: D(16 x 16) += A(16 x 4) B(4 x 16) with rows [w a | w res | 0 ...] and columns [a | res | 0 ...] of FOUR pixels per
//                   instruction (K = 4), 16 instructions per 64-pixel trip, the accumulator tile in 4 registers instead of 28.  A lane owns ONE pixel but
//                   the MFMA wants lane (i, k) to hold entry i of pixel k: the rows go through LDS (4 x ds_write_b128 per lane, 2 x ds_read_b32 + 2 selects
//                   per MFMA operand pair).
//
// Both accumulate the same sums (checked on the host up to fp32 summation order).  Output: one JSON line with the time per 64-pixel trip.
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_gram scripts/mfma_gram.hip && build/mfma_gram
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

// a pseudo pixel: seven values and a weight from one loaded float4 and the trip number (cheap, identical in both modes)
__device__ __forceinline__ void make_row(const float4 &s, int trip, float (&a)[6], float &res, float &w)
{
    const float t = (float)(trip & 15) * 0.0625f;
    a[0] = s.x + t; a[1] = s.y - t; a[2] = s.z * 0.5f; a[3] = s.w + 0.25f * t; a[4] = s.x * s.y; a[5] = s.z - s.w;
    res = 0.01f * (s.x - s.z) + 0.001f * t;
    w = fminf(1.0f, 0.005f * __builtin_amdgcn_rsqf(res * res + 1e-12f));
}

template <int MODE>
__global__ void __launch_bounds__(256, 6) k_gram(const float4 *__restrict__ in, float *__restrict__ out, int trips, unsigned long long *__restrict__ cycles)
{
    __shared__ __attribute__((aligned(16))) float stage[4][64][16];          // per wave: [pixel][w a (6), w res, 0 | a (6), res, 0]   (mode 1)
    __shared__ float red[4][28];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 s = in[(size_t)blockIdx.x * 256 + tid];
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
        float acc[28];
#pragma unroll
        for (int k = 0; k < 28; k++) acc[k] = 0.0f;
        for (int trip = 0; trip < trips; trip++) {
            float a[6], res, w;
            make_row(s, trip, a, res, w);
            f2v pr[6];
#pragma unroll
            for (int r = 0; r < 6; r++) { pr[r] = (f2v){ w * a[r], a[r] }; asm volatile("" : "+v"(pr[r])); }
            f2v rr = (f2v){ w, res };
            asm volatile("" : "+v"(rr));
            int k = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) {
#pragma unroll
                for (int c = r; c < 6; c++) acc[k++] += pr[r].x * pr[c].y;
                acc[21 + r] += pr[r].x * rr.y;
            }
            acc[27] += 1.0f;
        }
        // wave sums (plain butterflies: outside the timed loop's weight)
#pragma unroll
        for (int k = 0; k < 28; k++) {
            float v = acc[k];
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) red[wave][k] = v;
        }
    } else {
        f4v D = (f4v){ 0.f, 0.f, 0.f, 0.f };
        const int i16 = lane & 15, k4 = lane >> 4;           // MFMA operand coordinates of this lane: row / column i16, K index k4
        for (int trip = 0; trip < trips; trip++) {
            float a[6], res, w;
            make_row(s, trip, a, res, w);
            float4 *my = reinterpret_cast<float4 *>(&stage[wave][lane][0]);
            my[0] = make_float4(w * a[0], w * a[1], w * a[2], w * a[3]);
            my[1] = make_float4(w * a[4], w * a[5], w * res, 0.0f);
            my[2] = make_float4(a[0], a[1], a[2], a[3]);
            my[3] = make_float4(a[4], a[5], res, 1.0f);      // column 7: the count rides along as w-row x 1 ... (row 7 is zero: not used; the count is taken from D[6][7] = sum w res -- see host)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const float *p = &stage[wave][4 * g + k4][0];
                const float av = (i16 < 8) ? p[i16] : 0.0f;             // A[i][k] = (w row of pixel 4 g + k)[i]
                const float bv = (i16 < 8) ? p[8 + i16] : 0.0f;         // B[k][j] = (row of pixel 4 g + k)[j]
                D = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, D, 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // D[i][j]: lane (j = lane % 16, i = 4 * (lane / 16) + r) holds register r.  The 28 sums: S(r, c) r <= c < 6 = D[r][c], g_r = D[r][6] (w a_r res).
        if (lane == 0) for (int k = 0; k < 28; k++) red[wave][k] = 0.0f;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = 4 * k4 + r, j = i16;
            if (i < 6 && j >= i && j < 6) { int r0 = i, idx = r0 * 6 - r0 * (r0 - 1) / 2 + (j - r0); red[wave][idx] = D[r]; }
            if (i < 6 && j == 6) red[wave][21 + i] = D[r];
        }
        if (lane == 0) red[wave][27] = 64.0f * (float)trips;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (tid < 28) out[(size_t)blockIdx.x * 28 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    if (lane == 0) atomicAdd(cycles, t1 - t0);
}

int main()
{
    const int n_cu = 256, wg_per_cu = 6, trips = 2000;
    const int blocks = n_cu * wg_per_cu;
    std::vector<float4> h((size_t)blocks * 256);
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) / 16777216.0f - 0.5f; };
    for (auto &v : h) v = make_float4(rnd(), rnd(), rnd(), rnd());
    float4 *d_in; float *d_out[2]; unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_in, h.size() * sizeof(float4)));
    CHECK(hipMemcpy(d_in, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
    for (int m = 0; m < 2; m++) CHECK(hipMalloc(&d_out[m], (size_t)blocks * 28 * sizeof(float)));
    CHECK(hipMalloc(&d_cyc, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double ms[2], cyc[2];
    for (int m = 0; m < 2; m++) {
        for (int rep = 0; rep < 3; rep++) {          // the last repetition is the one reported
            CHECK(hipMemset(d_cyc, 0, 16));
            CHECK(hipEventRecord(e0));
            if (m == 0) k_gram<0><<<blocks, 256>>>(d_in, d_out[0], trips, d_cyc);
            else k_gram<1><<<blocks, 256>>>(d_in, d_out[1], trips, d_cyc);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms[m] = t;
            unsigned long long c; CHECK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
            cyc[m] = (double)c / ((double)blocks * 4.0 * trips);
        }
    }
    std::vector<float> o0((size_t)blocks * 28), o1((size_t)blocks * 28);
    CHECK(hipMemcpy(o0.data(), d_out[0], o0.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(o1.data(), d_out[1], o1.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    for (size_t k = 0; k < o0.size(); k++) { worst = std::fmax(worst, std::fabs((double)o0[k] - o1[k])); scale = std::fmax(scale, std::fabs((double)o0[k])); }
    const double trips_total = (double)blocks * 4.0 * trips;         // 64-pixel trips
    std::printf("{\"what\": \"Gram update w [a|res][a|res]^T of a 64-pixel trip, 6 workgroups of 256 threads per CU on 256 CUs, %d trips per wave\", "
                "\"valu\": {\"ms\": %.4f, \"ns_per_trip_per_simd\": %.3f, \"wave_cycles_per_trip\": %.1f}, "
                "\"mfma_16x16x4_f32_via_lds\": {\"ms\": %.4f, \"ns_per_trip_per_simd\": %.3f, \"wave_cycles_per_trip\": %.1f}, "
                "\"mfma_over_valu\": %.3f, \"max_abs_difference_of_the_sums\": %.4g, \"largest_sum\": %.4g}\n",
                trips, ms[0], ms[0] * 1e6 / (trips_total / 1024.0), cyc[0], ms[1], ms[1] * 1e6 / (trips_total / 1024.0), cyc[1], ms[1] / ms[0], worst, scale);
    return 0;
}
