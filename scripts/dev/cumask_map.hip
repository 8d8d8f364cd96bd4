// cumask_map.hip -- which compute units does a stream made by hipExtStreamCreateWithCUMask run on?  (developer probe, MI355X box)
//
// For a few masks: launch 8 192 one-wave workgroups that spin ~20 us each on the masked stream, every one records (XCC, SE, SH, CU) from
// HW_REG_XCC_ID / HW_REG_HW_ID; print how many distinct compute units were used and how they spread over the XCDs.  Answers: is bit i of
// the mask CU i of XCD (i mod 8) (the KFD's round-robin over XCCs), and can a stream be kept off a set of reserved CUs by the complement?
//   hipcc --offload-arch=gfx950 -O3 -o build/cumask_map scripts/dev/cumask_map.hip && build/cumask_map
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_where(unsigned *out, long long spin_ticks)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xFu) << 16) | (hw & 0xFF00u);       // xcc [19:16], se [15:13], sh [12], cu [11:8]
}

int main()
{
    const int blocks = 8192;
    unsigned *d;
    CHECK(hipMalloc(&d, blocks * 4));
    struct Case { const char *name; uint32_t w[8]; };
    const Case cases[] = {
        { "all 256 bits", { ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u } },
        { "bits 0-7", { 0xFFu, 0, 0, 0, 0, 0, 0, 0 } },
        { "bits 0-15", { 0xFFFFu, 0, 0, 0, 0, 0, 0, 0 } },
        { "bits 0-31", { ~0u, 0, 0, 0, 0, 0, 0, 0 } },
        { "bits 16-255 (complement of 0-15)", { 0xFFFF0000u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u } },
        { "bits 240-255", { 0, 0, 0, 0, 0, 0, 0, 0xFFFF0000u } },
        { "bit 0 and bit 8", { 0x101u, 0, 0, 0, 0, 0, 0, 0 } },
    };
    std::printf("[\n");
    bool first = true;
    for (const Case &c : cases) {
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, c.w);
        if (e != hipSuccess) { std::printf("%s {\"mask\": \"%s\", \"error\": \"%s\"}", first ? "" : ",\n", c.name, hipGetErrorString(e)); first = false; continue; }
        CHECK(hipMemsetAsync(d, 0xFF, blocks * 4, st));
        k_where<<<blocks, 64, 0, st>>>(d, 2000);             // 20 us at 100 MHz
        CHECK(hipStreamSynchronize(st));
        std::vector<unsigned> h(blocks);
        CHECK(hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost));
        std::set<unsigned> cus;
        std::map<unsigned, std::set<unsigned>> per_xcc;
        for (unsigned v : h) { cus.insert(v); per_xcc[v >> 16].insert(v & 0xFFFFu); }
        std::printf("%s {\"mask\": \"%s\", \"distinct_cus\": %zu, \"per_xcc\": {", first ? "" : ",\n", c.name, cus.size());
        bool f2 = true;
        for (auto &kv : per_xcc) {
            std::printf("%s\"%u\": [", f2 ? "" : ", ", kv.first); f2 = false;
            bool f3 = true;
            if (kv.second.size() > 8) std::printf("\"%zu compute units\"", kv.second.size());
            else for (unsigned v : kv.second) { std::printf("%s\"se%u.sh%u.cu%u\"", f3 ? "" : ", ", (v >> 13) & 7u, (v >> 12) & 1u, (v >> 8) & 15u); f3 = false; }
            std::printf("]");
        }
        std::printf("}}");
        first = false;
        CHECK(hipStreamDestroy(st));
    }
    std::printf("\n]\n");
    return 0;
}
