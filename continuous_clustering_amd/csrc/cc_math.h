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
    int id;
    float hi = 0.f, lo = 0.f;
    if (ix >= 0x4c000000) // |x| >= 2^25, inf, nan
    {
        if (ix > 0x7f800000)
            return x + x;
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    if (ix < 0x3ee00000) // |x| < 7/16
    {
        if (ix < 0x31000000) // |x| < 2^-29
            return x;
        id = -1;
    }
    else
    {
        x = absf(x);
        if (ix < 0x3f980000) // |x| < 19/16
        {
            if (ix < 0x3f300000) // 7/16 <= |x| < 11/16
            {
                id = 0;
                hi = hi0;
                lo = lo0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            }
            else
            {
                id = 1;
                hi = hi1;
                lo = lo1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        }
        else
        {
            if (ix < 0x401c0000) // |x| < 39/16
            {
                id = 2;
                hi = hi2;
                lo = lo2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            }
            else
            {
                id = 3;
                hi = hi3;
                lo = lo3;
                x = -1.0f / x;
            }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0)
        return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

// ---- atan2f --------------------------------------------------------------------------------
CCM_HD float atan2f_exact(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
                pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
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
    if (iy == 0x7f800000)
        return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else
        z = atanf_exact(absf(y / x));
    switch (m)
    {
        case 0:
            return z;
        case 1:
            return i2f(f2i(z) ^ (int32_t) 0x80000000);
        case 2:
            return pi - (z - pi_lo);
        default:
            return (z - pi_lo) - pi;
    }
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
    if (ix == 0x3f800000)
        return x * pio2_hi + x * pio2_lo;
    if (ix > 0x3f800000)
        return (x - x) / (x - x);
    if (ix < 0x3f000000) // |x| < 0.5
    {
        if (ix < 0x32000000) // |x| < 2^-27
            return x;
        const float t = x * x;
        const float w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
        return x + x * w;
    }
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
    return hx > 0 ? t : -t;
}

} // namespace ccm
