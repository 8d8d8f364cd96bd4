// btba_xorwow.hpp -- the reference's RANSAC sample stream, computed on the host.
//   reference: ransacEstimateModelKernel, src/cuda/cuda_ransac.cu:1154-1161 --
//       curandState state;  curand_init(0, idx, 0, &state);  rand_idx[k] = round(curand_uniform(&state) * (n_pts - 1)),  k = 0, 1, 2
//   one cuRAND XORWOW stream per trial idx, launched once per frame pair with the same seed: the three uniforms of trial idx do
//   not depend on the pair or on its points.  The whole "random" input of the reference's RANSAC is therefore ONE constant table
//   of n_trials x 3 floats; a pair turns row idx into its triple with one multiply and one round.  That is how it is built here:
//   the table is computed on the host (once per (seed, n_trials), kept in the workspace), uploaded (24 KB at 2 000 trials), and
//   k_ransac_vote reads three floats per lane -- instead of 2 000 curand_init calls per pair, each of which walks jump matrices
//   for its subsequence on the device (cuRAND's curand_init is the notoriously slow part of such kernels).
//
// cuRAND is CUDA-toolkit code (absent here); what follows implements its published XORWOW algorithm:
//   recurrence  -- Marsaglia, "Xorshift RNGs" (2003), xorwow: t = v0 ^ (v0 >> 2); v0..v3 = v1..v4;
//                  v4 = (v4 ^ (v4 << 4)) ^ (t ^ (t << 1)); d += 362437; output d + v4;
//   curand_init -- scramble (seed) into (v, d) as curand_kernel.h does (restated from the published header, UNVERIFIED: see seeded()), then advance by subsequence * 2^67 steps (+ offset steps);
//                  the five xorshift words evolve linearly over GF(2), so n steps are the 160 x 160 bit matrix A^n; the Weyl
//                  counter d moves by 362437 * n, which is 0 mod 2^32 for n = k 2^67;
//   uniform     -- x * 2^-32 + 2^-33 in fp32, in (0, 1].
// Formulation: bit matrices are kept by ROWS (output bit r = parity(row_r & state)); a product C = A B is built row by row as the
// XOR of B's rows selected by A's row.  J = A^(2^67) costs 67 squarings (~10 ms, once per process); trial t's state is J applied
// t times to the seeded state, one 160-parity mat-vec per trial.  The test oracle (oracle/xorwow.h) does the same arithmetic the
// other way round (column storage, XOR of columns) and checks both against rocRAND's precomputed jump tables.
#pragma once
#include <stdint.h>
#include <string.h>
#include <mutex>
#include <vector>

namespace btba {
namespace xorwow {

struct Bits160 { uint64_t w[3]; };                           // bits 0..159: word k of the generator = bits 32k .. 32k+31

inline Bits160 pack(const uint32_t v[5])
{
    Bits160 b;
    b.w[0] = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
    b.w[1] = (uint64_t)v[2] | ((uint64_t)v[3] << 32);
    b.w[2] = (uint64_t)v[4];
    return b;
}
inline void unpack(const Bits160 &b, uint32_t v[5])
{
    v[0] = (uint32_t)b.w[0]; v[1] = (uint32_t)(b.w[0] >> 32);
    v[2] = (uint32_t)b.w[1]; v[3] = (uint32_t)(b.w[1] >> 32);
    v[4] = (uint32_t)b.w[2];
}
struct Matrix { Bits160 row[160]; };                         // row r: which input bits feed output bit r

inline Bits160 apply(const Matrix &M, const Bits160 &x)
{
    Bits160 y = { { 0, 0, 0 } };
    for (int r = 0; r < 160; r++) {
        const uint64_t m = (M.row[r].w[0] & x.w[0]) ^ (M.row[r].w[1] & x.w[1]) ^ (M.row[r].w[2] & x.w[2]);
        y.w[r >> 6] |= (uint64_t)(__builtin_popcountll(m) & 1) << (r & 63);
    }
    return y;
}
inline void multiply(const Matrix &A, const Matrix &B, Matrix &out)          // out = A B (out may alias neither)
{
    for (int r = 0; r < 160; r++) {
        Bits160 acc = { { 0, 0, 0 } };
        for (int c = 0; c < 160; c++)
            if ((A.row[r].w[c >> 6] >> (c & 63)) & 1u) { acc.w[0] ^= B.row[c].w[0]; acc.w[1] ^= B.row[c].w[1]; acc.w[2] ^= B.row[c].w[2]; }
        out.row[r] = acc;
    }
}
// the recurrence's linear part, read off its definition: new v0..v3 = old v1..v4; new v4 = v4 ^ (v4 << 4) ^ t ^ (t << 1), t = v0 ^ (v0 >> 2)
inline void step_matrix(Matrix &A)
{
    memset(&A, 0, sizeof A);
    auto set = [&](int out_bit, int in_bit) { A.row[out_bit].w[in_bit >> 6] ^= 1ull << (in_bit & 63); };
    for (int k = 0; k < 4; k++)
        for (int b = 0; b < 32; b++) set(32 * k + b, 32 * (k + 1) + b);
    for (int b = 0; b < 32; b++) {
        const int o = 128 + b;
        set(o, 128 + b);                                     // v4
        if (b >= 4) set(o, 128 + b - 4);                     // v4 << 4
        set(o, b);                                           // t      = v0 ^ (v0 >> 2)
        if (b + 2 < 32) set(o, b + 2);
        if (b >= 1) {                                        // t << 1
            set(o, b - 1);
            if (b + 1 < 32) set(o, b + 1);
        }
    }
}
// A^(2^67): one subsequence
inline const Matrix &subsequence_jump()
{
    static Matrix J;
    static std::once_flag once;
    std::call_once(once, [] {
        Matrix a, b;
        step_matrix(a);
        for (int i = 0; i < 67; i++) { multiply(a, a, b); a = b; }
        J = a;
    });
    return J;
}

struct State { uint32_t d, v[5]; };

inline uint32_t next(State &s)                               // curand(&state)
{
    const uint32_t t = s.v[0] ^ (s.v[0] >> 2);
    s.v[0] = s.v[1]; s.v[1] = s.v[2]; s.v[2] = s.v[3]; s.v[3] = s.v[4];
    s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.d + s.v[4];
}
// Seed scrambling of curand_init, as published in curand_kernel.h (_curand_init_scratch): two salts XORed into the halves of the seed,
// two odd multipliers, then Marsaglia's five initial words and his d combined with the products as  +, ^, +, ^, +  and  d + t1 + t0.
// rocRAND's xorwow_engine (/opt/rocm/include/rocrand/rocrand_xorwow.h:113-122) is the same construction with four other constants, which
// is what lets a third-party implementation pin the operators here (tests/test_oracle_xorwow.py builds rocRAND's engine on the host and
// compares it with seeded(seed, kRocrandSeeding)).  The four cuRAND constants themselves rest on the published header alone:
// RESTATED, UNVERIFIED until a CUDA machine prints the rows INTEGRATION.md names.
struct SeedConstants { uint32_t salt_lo, salt_hi, mul_lo, mul_hi; };
constexpr SeedConstants kCurandSeeding = { 0xaad26b49u, 0xf7dcefddu, 1099087573u, 2591861531u };
constexpr SeedConstants kRocrandSeeding = { 0x2c7f967fu, 0xa03697cbu, 1228688033u, 2073658381u };
inline State seeded(uint64_t seed, const SeedConstants &k = kCurandSeeding)          // curand_init(seed, 0, 0)
{
    State s;
    const uint32_t s0 = (uint32_t)seed ^ k.salt_lo, s1 = (uint32_t)(seed >> 32) ^ k.salt_hi;
    const uint32_t t0 = k.mul_lo * s0, t1 = k.mul_hi * s1;
    s.d = 6615241u + t1 + t0;
    s.v[0] = 123456789u + t0; s.v[1] = 362436069u ^ t0; s.v[2] = 521288629u + t1; s.v[3] = 88675123u ^ t1; s.v[4] = 5783321u + t0;
    return s;
}
inline float uniform(State &s)                               // curand_uniform(&state)
{
#pragma clang fp contract(off)
    return (float)next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);      // CURAND_2POW32_INV; the product is exact
}

// u_out[3 t + k] = the k-th curand_uniform after curand_init(seed, t, 0), t = 0 .. n_trials-1 (raw_out, optional: the curand() words)
inline void ransac_uniform_table(uint64_t seed, int n_trials, float *u_out, const SeedConstants &k = kCurandSeeding, uint32_t *raw_out = nullptr)
{
    const Matrix &J = subsequence_jump();
    const State s0 = seeded(seed, k);
    Bits160 x = pack(s0.v);
    for (int t = 0; t < n_trials; t++) {
        State s;
        s.d = s0.d;                                          // a subsequence jump leaves the Weyl counter where it is
        unpack(x, s.v);
        for (int q = 0; q < 3; q++) {
            if (raw_out) { State c = s; raw_out[3 * t + q] = next(c); }
            u_out[3 * t + q] = uniform(s);
        }
        x = apply(J, x);
    }
}

}  // namespace xorwow
}  // namespace btba
