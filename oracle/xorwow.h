/* oracle/xorwow.h -- CPU restatement of cuRAND's XORWOW generator as the reference's RANSAC uses it:
 *     curandState state;  curand_init(0, idx, 0, &state);  round(curand_uniform(&state) * (n_pts - 1))   x 3
 * (/root/reference/src/cuda/cuda_ransac.cu:1154-1161, one stream per trial idx, the same for every frame pair).
 *
 * TEST INFRASTRUCTURE ONLY: included by oracle/btba_oracle_ransac.c and by oracle/ref_ransac_pre.h (where it stands in for
 * <curand_kernel.h> when the reference's own RANSAC kernels are compiled for the CPU).  Never part of the product path --
 * the product has its own implementation (bundletrack_amd/csrc/btba_xorwow.hpp, a different formulation: row-major bit
 * matrices and parities there, column XORs here), and tests compare the two bit for bit.
 *
 * THIRD-PARTY ALGORITHM, SOURCE ABSENT: cuRAND ships with the CUDA toolkit (version not pinned by the reference: its
 * Dockerfile takes whatever the nvidia/cudagl base image holds); neither the toolkit nor its headers exist in this image.
 * What is restated, and what each part is anchored on:
 *   - the recurrence: G. Marsaglia, "Xorshift RNGs", J. Stat. Software 8(14), 2003, section 3.1 `xorwow` -- five 32-bit
 *     xorshift words plus a Weyl counter d += 362437, output d + v[4].  Also identical to rocRAND's xorwow_engine::next()
 *     (/opt/rocm/include/rocrand/rocrand_xorwow.h), and tests/test_oracle_xorwow.py checks this file's one-step matrix
 *     against rocRAND's precomputed A^1;
 *   - curand_init(seed, subsequence, offset): seed scrambling as in curand_kernel.h's _curand_init_scratch (constants
 *     0xaad26b49, 0xf7dcefdd, 1099087573, 2591861531, combined with Marsaglia's initial words as +, ^, +, ^, + -- restated from
 *     the published header, NOT verifiable here: PARITY UNPINNED for the four constants; the operator pattern is the one
 *     rocRAND's cuRAND-derived engine uses and is checked against that engine with rocRAND's constants), then a jump of subsequence * 2^67 steps and of `offset` steps.  The 2^67 jump is
 *     the 160x160 GF(2) matrix A^(2^67) obtained here by 67 squarings; tests check it against rocRAND's precomputed
 *     h_xorwow_sequence_jump_matrices[0] (rocRAND documents the same 2^67 spacing).  d is unchanged by a subsequence jump
 *     (362437 * k * 2^67 = 0 mod 2^32) and advanced by 362437 * offset;
 *   - curand_uniform: x * 2^-32 + 2^-33 in fp32 (CURAND_2POW32_INV = 2.3283064e-10f; the product is exact, so contraction
 *     to an FMA cannot change it), result in (0, 1].
 * A real cuRAND run to compare with does not exist in this environment; DESIGN.md section 3 says so. */
#ifndef BTBA_ORACLE_XORWOW_H_
#define BTBA_ORACLE_XORWOW_H_
#include <stdint.h>
#include <string.h>

typedef struct { uint32_t d, v[5]; } orc_xorwow_state;

/* one step of Marsaglia's xorwow / curand(&state) */
static inline uint32_t orc_xorwow_next(orc_xorwow_state *s)
{
    const uint32_t t = s->v[0] ^ (s->v[0] >> 2);
    s->v[0] = s->v[1]; s->v[1] = s->v[2]; s->v[2] = s->v[3]; s->v[3] = s->v[4];
    s->v[4] = (s->v[4] ^ (s->v[4] << 4)) ^ (t ^ (t << 1));
    s->d += 362437u;
    return s->d + s->v[4];
}

/* A 160x160 matrix over GF(2) stored by COLUMNS: col[c] = the image (5 words) of unit vector c, c = 32 * word + bit. */
typedef struct { uint32_t col[160][5]; } orc_xorwow_mat;

static inline void orc_xorwow_matvec(const orc_xorwow_mat *M, const uint32_t in[5], uint32_t out[5])
{
    uint32_t r[5] = { 0, 0, 0, 0, 0 };
    for (int w = 0; w < 5; w++)
        for (int b = 0; b < 32; b++)
            if ((in[w] >> b) & 1u)
                for (int k = 0; k < 5; k++) r[k] ^= M->col[32 * w + b][k];
    memcpy(out, r, sizeof r);
}
/* the linear part of one step, as a matrix */
static inline void orc_xorwow_step_matrix(orc_xorwow_mat *A)
{
    for (int c = 0; c < 160; c++) {
        orc_xorwow_state s;
        memset(&s, 0, sizeof s);
        s.v[c / 32] = 1u << (c % 32);
        (void)orc_xorwow_next(&s);
        memcpy(A->col[c], s.v, sizeof s.v);
    }
}
static inline void orc_xorwow_mat_mul(const orc_xorwow_mat *A, const orc_xorwow_mat *B, orc_xorwow_mat *out)   /* out = A B */
{
    orc_xorwow_mat R;
    for (int c = 0; c < 160; c++) orc_xorwow_matvec(A, B->col[c], R.col[c]);
    *out = R;
}
/* A^(2^(base + k)), k = 0 .. 63, for base = 0 (plain steps) and base = 67 (subsequences): computed once, in order, and kept
 * (the reference calls curand_init once per trial and pair; the matrices are the same every time).  Not thread-safe: the
 * tests call this from one thread. */
static inline const orc_xorwow_mat *orc_xorwow_power(int which /* 0: base 0, 1: base 67 */, int k)
{
    static orc_xorwow_mat pw[2][64];
    static int have[2] = { 0, 0 };
    if (!have[which]) {
        orc_xorwow_step_matrix(&pw[which][0]);
        for (int i = 0; i < (which ? 67 : 0); i++) orc_xorwow_mat_mul(&pw[which][0], &pw[which][0], &pw[which][0]);
        have[which] = 1;
    }
    for (; have[which] <= k; have[which]++) orc_xorwow_mat_mul(&pw[which][have[which] - 1], &pw[which][have[which] - 1], &pw[which][have[which]]);
    return &pw[which][k];
}
/* state <- A^(n * 2^base) state, by the binary expansion of n (the Weyl counter is the caller's business) */
static inline void orc_xorwow_jump(orc_xorwow_state *s, uint64_t n, int which)
{
    for (int k = 0; n; n >>= 1, k++)
        if (n & 1u) orc_xorwow_matvec(orc_xorwow_power(which, k), s->v, s->v);
}

/* curand_init(seed, subsequence, offset, &state) for curandStateXORWOW_t, with the four seed-scrambling constants as a parameter:
 * consts = { salt of the low seed word, salt of the high word, multiplier of the low word, multiplier of the high word }.
 * Published curand_kernel.h (_curand_init_scratch): v[0] = 123456789 + t0, v[1] = 362436069 ^ t0, v[2] = 521288629 + t1,
 * v[3] = 88675123 ^ t1, v[4] = 5783321 + t0, d = 6615241 + t1 + t0.  rocRAND's xorwow_engine constructor
 * (/opt/rocm/include/rocrand/rocrand_xorwow.h:113-122) is the same construction with other constants -- with ITS constants this
 * function must reproduce rocRAND's engine bit for bit, which tests/test_oracle_xorwow.py checks against the engine itself. */
static const uint32_t orc_curand_seed_consts[4] = { 0xaad26b49u, 0xf7dcefddu, 1099087573u, 2591861531u };
static const uint32_t orc_rocrand_seed_consts[4] = { 0x2c7f967fu, 0xa03697cbu, 1228688033u, 2073658381u };
static inline void orc_xorwow_init_consts(const uint32_t consts[4], uint64_t seed, uint64_t subsequence, uint64_t offset, orc_xorwow_state *s)
{
    const uint32_t s0 = (uint32_t)seed ^ consts[0], s1 = (uint32_t)(seed >> 32) ^ consts[1];
    const uint32_t t0 = consts[2] * s0, t1 = consts[3] * s1;
    s->d = 6615241u + t1 + t0;
    s->v[0] = 123456789u + t0;
    s->v[1] = 362436069u ^ t0;
    s->v[2] = 521288629u + t1;
    s->v[3] = 88675123u ^ t1;
    s->v[4] = 5783321u + t0;
    orc_xorwow_jump(s, subsequence, 1);                  /* d: + 362437 * subsequence * 2^67 = + 0 (mod 2^32) */
    orc_xorwow_jump(s, offset, 0);
    s->d += 362437u * (uint32_t)offset;
}
static inline void orc_curand_init(uint64_t seed, uint64_t subsequence, uint64_t offset, orc_xorwow_state *s)
{
    orc_xorwow_init_consts(orc_curand_seed_consts, seed, subsequence, offset, s);
}
/* curand_uniform(&state): (0, 1] */
static inline float orc_curand_uniform(orc_xorwow_state *s)
{
    const uint32_t x = orc_xorwow_next(s);
    return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
#endif
