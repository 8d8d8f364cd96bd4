// xorwow_vs_rocrand.cpp -- holds both restatements of the XORWOW sample stream against THIRD-PARTY code: rocRAND's own
// xorwow_engine (/opt/rocm/include/rocrand/rocrand_xorwow.h, a __host__ __device__ header), run here on the host.
//
// cuRAND (what the reference's RANSAC draws from, /root/reference/src/cuda/cuda_ransac.cu:1154-1161) is absent from this image;
// rocRAND's engine is cuRAND-derived: same recurrence, same 2^67 subsequence spacing, same seed-scrambling CONSTRUCTION
// (+, ^, +, ^, + on Marsaglia's initial words, d + t1 + t0) with four other constants.  So with rocRAND's constants substituted
//   * the product's generator  (bundletrack_amd/csrc/btba_xorwow.hpp: seeded(), the 67-squarings jump matrix, next())  and
//   * the oracle's generator   (oracle/xorwow.h: orc_xorwow_init_consts(), column-stored jump powers, orc_xorwow_next())
// must reproduce rocRAND's engine bit for bit: state after (seed, subsequence, offset) and the raw draws.  That pins the
// operators of the seeding, the subsequence jump, the offset jump and the recurrence against code neither of them shares.
// What it cannot pin: cuRAND's four constants and its uniform conversion (rocRAND's is x 2^-32 + 2^-32, cuRAND's published
// one x 2^-32 + 2^-33).
//
// Built and run by tests/test_oracle_xorwow.py (g++, host only, no GPU).  Prints one line per check and "ALL OK" at the end.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <rocrand/rocrand_xorwow.h>
#include "../../bundletrack_amd/csrc/btba_xorwow.hpp"
extern "C" {
#include "../../oracle/xorwow.h"
}

namespace {
struct Peek : rocrand_device::xorwow_engine {                  // the engine's state is a protected member
    Peek(unsigned long long seed, unsigned long long sub, unsigned long long off) : rocrand_device::xorwow_engine(seed, sub, off) {}
    unsigned d() const { return m_state.d; }
    const unsigned *x() const { return m_state.x; }
};
}

int main()
{
    using namespace btba::xorwow;
    int bad = 0;
    const unsigned long long seeds[] = { 0ull, 1ull, 17ull, 1234567ull, (1ull << 40) + 5ull, ~0ull };
    const unsigned long long subs[] = { 0ull, 1ull, 2ull, 3ull, 7ull, 64ull, 1999ull, 0xDEADBEEF12345ull };
    const Matrix &J = subsequence_jump();
    for (unsigned long long seed : seeds) {
        // product: seeded state, then J applied t times for t = 0 .. 1999 (how ransac_uniform_table walks the trials)
        const State s0 = seeded(seed, kRocrandSeeding);
        Bits160 xs = pack(s0.v);
        for (unsigned long long t = 0; t < 2000; t++) {
            if (t < 40 || t % 97 == 0 || t == 1999) {
                Peek e(seed, t, 0);
                State s; s.d = s0.d; unpack(xs, s.v);
                bool ok = s.d == e.d() && std::memcmp(s.v, e.x(), sizeof s.v) == 0;
                for (int k = 0; k < 3; k++) ok = ok && next(s) == e.next();
                if (!ok) { std::printf("FAIL product seed %llu subsequence %llu\n", seed, t); bad++; }
            }
            xs = apply(J, xs);
        }
        // the table entry point itself (raw words) against the engine
        {
            const int n = 64;
            float u[3 * n]; uint32_t raw[3 * n];
            ransac_uniform_table(seed, n, u, kRocrandSeeding, raw);
            for (int t = 0; t < n; t++) {
                Peek e(seed, (unsigned long long)t, 0);
                for (int k = 0; k < 3; k++) {
                    const unsigned w = e.next();
                    if (raw[3 * t + k] != w) { std::printf("FAIL product table seed %llu trial %d draw %d\n", seed, t, k); bad++; }
#pragma clang fp contract(off)
                    const float want = (float)w * 2.3283064e-10f + (2.3283064e-10f / 2.0f);       // the published cuRAND conversion of the same word
                    if (std::memcmp(&want, &u[3 * t + k], 4) != 0) { std::printf("FAIL product uniform seed %llu trial %d draw %d\n", seed, t, k); bad++; }
                }
            }
        }
        // oracle: arbitrary subsequences and offsets through its binary-expansion jump
        for (unsigned long long sub : subs)
            for (unsigned long long off : { 0ull, 1ull, 5ull, 4099ull }) {
                Peek e(seed, sub, off);
                orc_xorwow_state s;
                orc_xorwow_init_consts(orc_rocrand_seed_consts, seed, sub, off, &s);
                bool ok = s.d == e.d() && std::memcmp(s.v, e.x(), sizeof s.v) == 0;
                for (int k = 0; k < 5; k++) ok = ok && orc_xorwow_next(&s) == e.next();
                if (!ok) { std::printf("FAIL oracle seed %llu subsequence %llu offset %llu\n", seed, sub, off); bad++; }
            }
        std::printf("seed %llu checked\n", seed);
    }
    // the default (cuRAND) constants differ from rocRAND's in nothing but the four numbers
    {
        const State a = seeded(0), b = seeded(0, kCurandSeeding);
        if (std::memcmp(&a, &b, sizeof a) != 0) { std::printf("FAIL default constants\n"); bad++; }
        orc_xorwow_state s;
        orc_curand_init(0, 0, 0, &s);
        if (s.d != a.d || std::memcmp(s.v, a.v, sizeof s.v) != 0) { std::printf("FAIL product / oracle seeded states differ\n"); bad++; }
        std::printf("curand_init(0, 0, 0) restated: d = %u, v = %u %u %u %u %u\n", a.d, a.v[0], a.v[1], a.v[2], a.v[3], a.v[4]);
    }
    if (!bad) std::printf("ALL OK\n");
    return bad ? 1 : 0;
}
