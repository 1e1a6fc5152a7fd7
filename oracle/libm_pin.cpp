// libm_pin.cpp — TEST INFRASTRUCTURE (not shipped, not on the product path).
//
// Pins continuous_clustering_amd/csrc/cc_math.h against the libm the reference links
// (glibc 2.35, /lib/x86_64-linux-gnu/libm.so.6 in this image): the reference calls
// std::atan2(float,float) at continuous_clustering.cpp:142 and :598 and std::asin(float) at
// :232 and :805.
//
//   libm_pin full     asinf + atanf over all 2^32 bit patterns, atan2f over 2^31 structured+random pairs
//   libm_pin quick    2^24 strided inputs per function (used by the CPU test-suite)
//
// Prints "mismatches: N" per function; exit code 0 iff all are 0. NaN results compare equal
// when both are NaN (payload/sign of a NaN result never reaches a parity output).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

#include "../continuous_clustering_amd/csrc/cc_math.h"

static inline bool same(float a, float b)
{
    if (std::isnan(a) && std::isnan(b))
        return true;
    return ccm::f2i(a) == ccm::f2i(b);
}

template<class F>
static uint64_t par(uint64_t n, F f)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0)
        nt = 4;
    std::atomic<uint64_t> bad{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back(
            [&, t]()
            {
                uint64_t b = 0;
                for (uint64_t i = t; i < n; i += nt)
                    b += f(i);
                bad += b;
            });
    for (auto& x : th)
        x.join();
    return bad.load();
}

static inline uint64_t mix(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

int main(int argc, char** argv)
{
    bool full = argc > 1 && !strcmp(argv[1], "full");
    uint64_t n1 = full ? (1ull << 32) : (1ull << 24);
    uint64_t stride = full ? 1 : 251; // odd stride walks all exponent ranges in quick mode
    volatile float sink = 0;
    (void) sink;

    uint64_t bad_asin = par(n1,
                            [&](uint64_t i) -> uint64_t
                            {
                                uint32_t u = (uint32_t) (i * stride + (full ? 0 : i * 7919));
                                float x = ccm::i2f((int32_t) u);
                                float a = asinf(x), b = ccm::asinf_exact(x);
                                if (!same(a, b))
                                {
                                    return 1;
                                }
                                return 0;
                            });
    printf("asinf  inputs %llu mismatches: %llu\n", (unsigned long long) n1, (unsigned long long) bad_asin);

    uint64_t bad_atan = par(n1,
                            [&](uint64_t i) -> uint64_t
                            {
                                uint32_t u = (uint32_t) (i * stride + (full ? 0 : i * 7919));
                                float x = ccm::i2f((int32_t) u);
                                return same(atanf(x), ccm::atanf_exact(x)) ? 0 : 1;
                            });
    printf("atanf  inputs %llu mismatches: %llu\n", (unsigned long long) n1, (unsigned long long) bad_atan);

    // atan2f: (a) random bit patterns for both operands, (b) lidar-like magnitudes with random signs,
    // (c) the special grid {0, -0, 1, -1, inf, -inf, nan, tiny, huge} x same.
    uint64_t n2 = full ? (1ull << 31) : (1ull << 23);
    uint64_t bad_atan2 = par(n2,
                             [&](uint64_t i) -> uint64_t
                             {
                                 uint64_t r = mix(i);
                                 float y, x;
                                 if (i & 1)
                                 {
                                     y = ccm::i2f((int32_t) (r & 0xffffffffu));
                                     x = ccm::i2f((int32_t) (r >> 32));
                                 }
                                 else
                                 {
                                     // |v| in [2^-6, 2^8) with full mantissa entropy
                                     uint32_t my = (uint32_t) (r & 0x7fffff), mx = (uint32_t) ((r >> 23) & 0x7fffff);
                                     uint32_t ey = 121 + (uint32_t) ((r >> 46) % 14), ex = 121 + (uint32_t) ((r >> 50) % 14);
                                     uint32_t sy = (uint32_t) ((r >> 62) & 1), sx = (uint32_t) ((r >> 63) & 1);
                                     y = ccm::i2f((int32_t) ((sy << 31) | (ey << 23) | my));
                                     x = ccm::i2f((int32_t) ((sx << 31) | (ex << 23) | mx));
                                 }
                                 return same(atan2f(y, x), ccm::atan2f_exact(y, x)) ? 0 : 1;
                             });
    const float sp[] = {0.f,     -0.f,    1.f,      -1.f,     INFINITY, -INFINITY, NAN,    1e-40f, -1e-40f,
                        1e38f,   -1e38f,  0.5f,     -0.5f,    2.f,      -2.f,      1e-30f, 1e30f,  3.14159274f,
                        0.4375f, 0.6875f, 1.1875f,  2.4375f,  3.4e38f,  1.17549435e-38f};
    for (float y : sp)
        for (float x : sp)
            if (!same(atan2f(y, x), ccm::atan2f_exact(y, x)))
                bad_atan2++;
    // the ignore test of the reference: atan2f(max_distance, distance), distance sweeping all floats in [0.01, 400)
    for (float md : {0.5f, 0.7f, 0.3f, 1.0f})
        for (float d = 0.01f; d < 400.f; d = std::nextafterf(d, 1e9f))
            if (!same(atan2f(md, d), ccm::atan2f_exact(md, d)))
                bad_atan2++;
    printf("atan2f pairs  %llu (+grid, +ignore sweep) mismatches: %llu\n", (unsigned long long) n2,
           (unsigned long long) bad_atan2);
    return (bad_asin | bad_atan | bad_atan2) ? 1 : 0;
}
