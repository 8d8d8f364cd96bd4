// btba_image.hpp -- the step just before the optimiser boundary: depth
// pre-processing and normals, i.e. what Frame's constructor runs on every incoming frame
//   Frame::processDepth            src/Frame.cpp:152-180  -> erodeDepthMapDevice  CUDAImageUtil.cu:676-718
//                                                            gaussFilterDepthMapDevice x2  :735-797
//   Frame::depthToCloudAndNormals  src/Frame.cpp:182-233  -> convertDepthFloatToCameraSpaceFloat4 :310-327
//                                                            computeNormals_Kernel :342-412
// The reference issues 3 + 2 launches with four full-image round trips through device memory; here each
// group is ONE launch: the erode -> filter -> filter chain runs on an LDS tile with a (r_e + 2 r_f)-pixel
// halo, and the normals kernel recomputes the five camera-space points it needs from depth instead of
// materialising a float4 xyz map.
#pragma once
#include <hip/hip_runtime.h>
#include "btba_device.hpp"

namespace btba {

constexpr int kTileW = 32, kTileH = 8;          // 256 output pixels per workgroup

struct DepthFilterParams {
    int W, H;
    int erode_radius; float erode_diff, erode_ratio;
    int bf_radius; float sigma_d, sigma_r;
};

// dynamic LDS: two float planes of (kTileW + 2h) x (kTileH + 2h), h = erode_radius + 2 bf_radius.
// RE / RF: the radii at compile time (the tracker's 1 and 2: 3 x 3 and 5 x 5 windows fully unrolled, LDS reads at immediate offsets), or -1 for
// radii read from P.  A workgroup whose whole LDS tile lies inside the image (INTERIOR: all but the frame's rim) drops the four coordinate
// tests per tap.  The arithmetic per tap, for every variant:
//   * the mean gate `(double)|c - mean| < 0.01` of the reference is decided as `|c - mean| <= 0.01f`: the float nearest 0.01 lies below
//     the double 0.01 and its successor above, so the two tests accept the same floats;
//   * the weight is exp(s(dx, dy) - (centre - c)^2 / (2 sigma_r^2)) with the spatial term s = -(dx^2 + dy^2) / (2 sigma_d^2) divided once per
//     thread (it takes six values for a 5 x 5 window) instead of once per tap, and the range term multiplied by 1 / (2 sigma_r^2).
template <int RE, int RF, bool INTERIOR>
__device__ __forceinline__ void process_depth_tile(const DepthFilterParams &P, const float *__restrict__ in, float *__restrict__ out, float *tile)
{
    const int re = RE >= 0 ? RE : P.erode_radius, rf = RF >= 0 ? RF : P.bf_radius, h = re + 2 * rf;
    const int LW = kTileW + 2 * h, LH = kTileH + 2 * h;
    float *b0 = tile, *b1 = tile + LW * LH;
    const int x0 = blockIdx.x * kTileW - h, y0 = blockIdx.y * kTileH - h;       // image coords of LDS (0,0)
    const int tid = threadIdx.x;
    auto inside = [&](int gx, int gy) { return INTERIOR || (gx >= 0 && gx < P.W && gy >= 0 && gy < P.H); };
    // stage 0: raw depth (anything for out-of-image cells: consumers test coordinates, not values)
    for (int e = tid; e < LW * LH; e += 256) {
        const int lx = e % LW, ly = e / LW, gx = x0 + lx, gy = y0 + ly;
        b0[e] = inside(gx, gy) ? in[(size_t)gy * P.W + gx] : 0.0f;
    }
    __syncthreads();
    // stage 1: erode on the region that the two filter passes will read (margin re)
    {
        const int m = re, RW = LW - 2 * m, RH = LH - 2 * m;
        const unsigned win = (2 * re + 1) * (2 * re + 1);
        for (int e = tid; e < RW * RH; e += 256) {
            const int lx = m + e % RW, ly = m + e / RW, gx = x0 + lx, gy = y0 + ly;
            float o = 0.0f;
            if (inside(gx, gy)) {
                const float old = b0[ly * LW + lx];
                if (!(old <= 0.1f)) {
                    unsigned count = 0;
#pragma unroll
                    for (int i = -re; i <= re; i++)
#pragma unroll
                        for (int j = -re; j <= re; j++)
                            if (inside(gx + j, gy + i)) {
                                const float d = b0[(ly + i) * LW + lx + j];
                                if (d == -INFINITY || d < 0.1f || fabsf(d - old) > P.erode_diff) count++;
                            }
                    o = ((float)count / (float)win >= P.erode_ratio) ? 0.0f : old;
                }
            }
            b1[ly * LW + lx] = o;
        }
    }
    __syncthreads();
    // stages 2 and 3: the mean-gated bilateral filter, twice
    const float two_sd2 = 2.0f * P.sigma_d * P.sigma_d, two_sr2 = 2 * P.sigma_r * P.sigma_r;
    const float num_total = (float)((2 * rf + 1) * (2 * rf + 1));
    const float inv_two_sr2 = 1.0f / two_sr2;             // (the reference divides per tap: one rounding of a term that is <= 0.5 and usually ~1e-15)
    for (int pass = 0; pass < 2; pass++) {
        const float *src = pass == 0 ? b1 : b0;
        float *dst = pass == 0 ? b0 : nullptr;
        const int m = re + rf * (pass + 1), RW = LW - 2 * m, RH = LH - 2 * m;
        for (int e = tid; e < RW * RH; e += 256) {
            const int lx = m + e % RW, ly = m + e / RW, gx = x0 + lx, gy = y0 + ly;
            if (!inside(gx, gy)) continue;
            float o = 0.0f;
            const float *ctr = src + ly * LW + lx;
            const float centre = ctr[0];
            float mean = 0.0f;
            int nvalid = 0;
#pragma unroll
            for (int dx = -rf; dx <= rf; dx++)                 // same nesting as the reference: x outer, y inner
#pragma unroll
                for (int dy = -rf; dy <= rf; dy++)
                    if (inside(gx + dx, gy + dy)) {
                        const float c = ctr[dy * LW + dx];
                        if (c >= 0.1f) { nvalid++; mean += c; }
                    }
            if (nvalid > 0) {
                mean /= (float)nvalid;
                float sum = 0.0f, sw = 0.0f;
#pragma unroll
                for (int dx = -rf; dx <= rf; dx++)
#pragma unroll
                    for (int dy = -rf; dy <= rf; dy++)
                        if (inside(gx + dx, gy + dy)) {
                            const float c = ctr[dy * LW + dx];
                            if (c >= 0.1f && fabsf(c - mean) <= 0.01f) {
                                const float spatial = -(float)(dx * dx + dy * dy) / two_sd2;          // (compile-time dx, dy: hoisted out of the pixel loop)
                                const float wgt = __expf(spatial - (centre - c) * (centre - c) * inv_two_sr2);       // arguments in [-(2 rf^2) / (2 sigma_d^2) - ..., 0]: no range handling needed
                                sw += wgt;
                                sum += wgt * c;
                            }
                        }
                if (sw > 0.0f && (float)nvalid / num_total > 0) o = sum / sw;
            }
            if (dst) dst[ly * LW + lx] = o;
            else out[(size_t)gy * P.W + gx] = o;
        }
        __syncthreads();
    }
}

template <int RE, int RF>
__global__ void __launch_bounds__(256) k_process_depth(DepthFilterParams P, const float *__restrict__ in, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int h = (RE >= 0 ? RE : P.erode_radius) + 2 * (RF >= 0 ? RF : P.bf_radius);
    const int x0 = blockIdx.x * kTileW - h, y0 = blockIdx.y * kTileH - h;
    const bool interior = x0 >= 0 && y0 >= 0 && x0 + kTileW + 2 * h <= P.W && y0 + kTileH + 2 * h <= P.H;      // (uniform)
    if (interior) process_depth_tile<RE, RF, true>(P, in, out, tile);
    else process_depth_tile<RE, RF, false>(P, in, out, tile);
}

// camera-space point of pixel (x, y): intrinsicsInv * (x d, y d, d, d), z = d, zeros when d < 0.1
__device__ __forceinline__ float3 backproject(const float *Ki, int x, int y, float d)
{
#pragma clang fp contract(off)
    if (!((double)d >= 0.1)) return make_float3(0.f, 0.f, 0.f);
    const float vx = (float)x * d, vy = (float)y * d;
    return make_float3(Ki[0] * vx + Ki[1] * vy + Ki[2] * d + Ki[3] * d, Ki[4] * vx + Ki[5] * vy + Ki[6] * d + Ki[7] * d, Ki[12] * vx + Ki[13] * vy + Ki[14] * d + Ki[15] * d);
}

// grid (ceil(W/64), ceil(H/4)) x (64, 4).  normals (and optionally the xyz map) for one frame.
__global__ void __launch_bounds__(256) k_depth_to_normals(int W, int H, Mat4 Kinv, const float *__restrict__ depth, float4 *__restrict__ normals, float4 *__restrict__ xyz_out)
{
#pragma clang fp contract(off)
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const size_t o = (size_t)y * W + x;
    const float dC = depth[o];
    const float3 CC = backproject(Kinv.m, x, y, dC);
    if (xyz_out) xyz_out[o] = ((double)dC >= 0.1) ? make_float4(CC.x, CC.y, CC.z, 1.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
    const float thr = 0.02f;
    if (x > 0 && x < W - 1 && y > 0 && y < H - 1 && !(CC.z < 0.1f)) {
        const float3 PC = backproject(Kinv.m, x, y + 1, depth[o + W]), MC = backproject(Kinv.m, x, y - 1, depth[o - W]);
        const float3 CP = backproject(Kinv.m, x + 1, y, depth[o + 1]), CM = backproject(Kinv.m, x - 1, y, depth[o - 1]);
        float3 xd, yd;
        bool ok = true;
        if (PC.z >= 0.1f && MC.z >= 0.1f && fabsf(PC.z - CC.z) <= thr && fabsf(MC.z - CC.z) <= thr) xd = make_float3(PC.x - MC.x, PC.y - MC.y, PC.z - MC.z);
        else if (PC.z >= 0.1f && fabsf(PC.z - CC.z) <= thr) xd = make_float3(PC.x - CC.x, PC.y - CC.y, PC.z - CC.z);
        else if (MC.z >= 0.1f && fabsf(MC.z - CC.z) <= thr) xd = make_float3(MC.x - CC.x, MC.y - CC.y, MC.z - CC.z);
        else ok = false;
        if (CP.z >= 0.1f && CM.z >= 0.1f && fabsf(CP.z - CC.z) <= thr && fabsf(CM.z - CC.z) <= thr) yd = make_float3(CP.x - CM.x, CP.y - CM.y, CP.z - CM.z);
        else if (CP.z >= 0.1f && fabsf(CP.z - CC.z) <= thr) yd = make_float3(CP.x - CC.x, CP.y - CC.y, CP.z - CC.z);
        else if (CM.z >= 0.1f && fabsf(CM.z - CC.z) <= thr) yd = make_float3(CM.x - CC.x, CM.y - CC.y, CM.z - CC.z);
        else ok = false;
        if (ok) {
            float nx = xd.y * yd.z - xd.z * yd.y, ny = xd.z * yd.x - xd.x * yd.z, nz = xd.x * yd.y - xd.y * yd.x;
            const float l = sqrtf(nx * nx + ny * ny + nz * nz);
            nx = nx / l; ny = ny / l; nz = nz / l;
            if (nx * -CC.x + ny * -CC.y + nz * -CC.z < 0) { nx = -nx; ny = -ny; nz = -nz; }
            if (l > 0.0f) res = make_float4(nx, ny, nz, 0.0f);
        }
    }
    normals[o] = res;
}

}  // namespace btba
