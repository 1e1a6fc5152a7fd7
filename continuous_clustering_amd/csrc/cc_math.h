// cc_math.h — bit-exact single-precision atanf / atan2f / asinf for host and gfx950 device.
//
// Why this exists: the reference decides range-image columns and ignore flags with glibc's
// atan2f / asinf (src/clustering/continuous_clustering.cpp:142, :232, :598, :805). A 1-ulp
// difference can move a point into the neighbouring column, so the HIP kernels cannot use
// ROCm's OCML versions. These are restatements of the published fdlibm/glibc-2.35 float
// algorithms (sysdeps/ieee754/flt-32/{s_atanf,e_atan2f,e_asinf}.c, glibc 2.35 is the libm of
// the reference's runtime image and of this image) using only IEEE + - * / sqrt, no FMA.
// oracle/libm_pin.c checks them against the container's libm.so.6: asinf and atanf over all
// 2^32 inputs, atan2f over structured + random pairs. Build device code with
// -ffp-contract=off so that none of the a*b+c expressions below is fused.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CCM_HD __host__ __device__ __forceinline__
#else
#define CCM_HD static inline
#endif

namespace ccm
{

CCM_HD int32_t f2i(float f)
{
    return __builtin_bit_cast(int32_t, f);
}
CCM_HD float i2f(int32_t i)
{
    return __builtin_bit_cast(float, i);
}
CCM_HD float absf(float x)
{
    return i2f(f2i(x) & 0x7fffffff);
}
// correctly rounded sqrt on both sides (hipcc: -fhip-fp32-correctly-rounded-divide-sqrt is the default)
CCM_HD float sqrt_rn(float x)
{
    return __builtin_sqrtf(x);
}

// Round 4: the three functions are written with selects instead of the nested ifs of the C sources. The values computed are the same
// (same operations on the same operands in the same order; the exhaustive comparison with libm.so.6 was repeated: oracle/libm_pin_full.log) —
// but on the GPU a divergent `if` is a pair of exec-mask instructions plus a skip branch, and a wavefront that is alone on its SIMD pays
// 15 - 30 clocks for each of them (lone-wave cost model): fdlibm's five argument ranges with a division each became one division.
// Rare cases (NaN / infinite / zero arguments, |x| >= 0.5 in asinf) sit behind ONE test; on the device that test is wave-uniform.
#if defined(__HIP_DEVICE_COMPILE__)
#define CCM_ANY(c) __any(c)
#else
#define CCM_ANY(c) (c)
#endif

// ---- atanf ---------------------------------------------------------------------------------
CCM_HD float atanf_exact(float x)
{
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f,
                hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f,
                lo3 = 7.5497894159e-08f;
    const float a0 = 3.3333334327e-01f, a1 = -2.0000000298e-01f, a2 = 1.4285714924e-01f,
                a3 = -1.1111110449e-01f, a4 = 9.0908870101e-02f, a5 = -7.6918758452e-02f,
                a6 = 6.6610731184e-02f, a7 = -5.8335702866e-02f, a8 = 4.9768779427e-02f,
                a9 = -3.6531571299e-02f, a10 = 1.6285819933e-02f;
    const int32_t hx = f2i(x);
    const int32_t ix = hx & 0x7fffffff;
    const float ax = absf(x);
    // argument reduction: id -1: |x| < 7/16 (t = x), 0: < 11/16, 1: < 19/16, 2: < 39/16, 3: above; t = num / den (x / 1 is x)
    const bool r0 = ix < 0x3ee00000, r1 = ix < 0x3f300000, r2 = ix < 0x3f980000, r3 = ix < 0x401c0000;
    const float num = r0 ? x : (r1 ? 2.0f * ax - 1.0f : (r2 ? ax - 1.0f : (r3 ? ax - 1.5f : -1.0f)));
    const float den = r0 ? 1.0f : (r1 ? 2.0f + ax : (r2 ? ax + 1.0f : (r3 ? 1.0f + 1.5f * ax : ax)));
    const float hi = r1 ? hi0 : (r2 ? hi1 : (r3 ? hi2 : hi3));
    const float lo = r1 ? lo0 : (r2 ? lo1 : (r3 ? lo2 : lo3));
    const float t = num / den;
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    const float small = t - t * (s1 + s2);
    const float r = hi - ((t * (s1 + s2) - lo) - t);
    float res = r0 ? small : (hx < 0 ? -r : r);
    res = ix < 0x31000000 ? x : res; // |x| < 2^-29
    if (CCM_ANY(ix >= 0x4c000000))   // |x| >= 2^25, inf, nan
    {
        if (ix >= 0x4c000000)
            res = ix > 0x7f800000 ? x + x : (hx > 0 ? hi3 + lo3 : -hi3 - lo3);
    }
    return res;
}

// ---- atan2f --------------------------------------------------------------------------------
// the cases fdlibm decides before it divides: NaN, x == 1, zeros, infinities
CCM_HD float atan2f_special(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f;
    const int32_t hx = f2i(x), hy = f2i(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000)
        return x + y;
    if (hx == 0x3f800000)
        return atanf_exact(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2); // 2*sign(x) + sign(y)
    if (iy == 0)
    {
        if (m < 2)
            return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0)
        return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000)
    {
        if (iy == 0x7f800000)
        {
            switch (m)
            {
                case 0:
                    return pi_o_4 + tiny;
                case 1:
                    return -pi_o_4 - tiny;
                case 2:
                    return 3.0f * pi_o_4 + tiny;
                default:
                    return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m)
        {
            case 0:
                return 0.0f;
            case 1:
                return -0.0f;
            case 2:
                return pi + tiny;
            default:
                return -pi - tiny;
        }
    }
    return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny; // iy == 0x7f800000
}

CCM_HD float atan2f_exact(float y, float x)
{
    const float pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = f2i(x), hy = f2i(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    const bool special = (ix >= 0x7f800000) | (iy >= 0x7f800000) | (hx == 0x3f800000) | (iy == 0) | (ix == 0);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2); // 2*sign(x) + sign(y)
    const int32_t k = (iy - ix) >> 23;
    const float za = atanf_exact(absf(y / x));
    const float z = k > 60 ? pi_o_2 + 0.5f * pi_lo : ((hx < 0 && k < -60) ? 0.0f : za);
    const float zz = z - pi_lo;
    float res = m == 0 ? z : (m == 1 ? i2f(f2i(z) ^ (int32_t) 0x80000000) : (m == 2 ? pi - zz : zz - pi));
    if (CCM_ANY(special))
    {
        if (special)
            res = atan2f_special(y, x);
    }
    return res;
}

// ---- asinf ---------------------------------------------------------------------------------
CCM_HD float asinf_exact(float x)
{
    const float pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f,
                pio4_hi = 0.785398185253143310546875f;
    const float p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f,
                p4 = 4.216630880e-2f;
    const int32_t hx = f2i(x);
    const int32_t ix = hx & 0x7fffffff;
    // |x| < 0.5 (every laser of a rotating sensor: inclinations within 30 degrees of the horizon)
    const float t0 = x * x;
    const float w0 = t0 * (p0 + t0 * (p1 + t0 * (p2 + t0 * (p3 + t0 * p4))));
    float res = ix < 0x32000000 ? x : x + x * w0; // |x| < 2^-27: x
    if (CCM_ANY(ix >= 0x3f000000))
    {
        if (ix >= 0x3f000000)
        {
            if (ix == 0x3f800000)
                res = x * pio2_hi + x * pio2_lo;
            else if (ix > 0x3f800000)
                res = (x - x) / (x - x);
            else
            {
                float w = 1.0f - absf(x);
                float t = w * 0.5f;
                float p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
                const float s = sqrt_rn(t);
                if (ix >= 0x3F79999A) // |x| > 0.975
                {
                    t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
                }
                else
                {
                    w = i2f(f2i(s) & (int32_t) 0xfffff000);
                    const float c = (t - w * w) / (s + w);
                    const float r = p;
                    p = 2.0f * s * r - (pio2_lo - 2.0f * c);
                    const float q = pio4_hi - 2.0f * w;
                    t = pio4_hi - (p - q);
                }
                res = hx > 0 ? t : -t;
            }
        }
    }
    return res;
}

} // namespace ccm
