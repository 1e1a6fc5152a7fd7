// math_probe.hip — TEST INFRASTRUCTURE: evaluates the device versions of the bit-exact math used by the kernels
// (cc_math.h + the f64 sqrt / f32 divide the insertion kernel relies on) so that tests can compare them with the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../continuous_clustering_amd/csrc/cc_math.h"

#pragma clang fp contract(off)

__global__ void k_probe(int n, const float* a, const float* b, const double* d, float* o_atan2, float* o_asin, float* o_div,
                        float* o_sqrtf, double* o_sqrtd, float* o_len)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    o_atan2[i] = ccm::atan2f_exact(a[i], b[i]);
    o_asin[i] = ccm::asinf_exact(a[i]);
    o_div[i] = a[i] / b[i];
    o_sqrtf[i] = ccm::sqrt_rn(ccm::absf(a[i]));
    o_sqrtd[i] = __builtin_sqrt(d[i]);
    o_len[i] = ccm::sqrt_rn(a[i] * a[i] + b[i] * b[i]);
}

extern "C" int math_probe(int n, const float* a, const float* b, const double* d, float* o_atan2, float* o_asin, float* o_div,
                          float* o_sqrtf, double* o_sqrtd, float* o_len)
{
    float *da, *db, *o1, *o2, *o3, *o4, *o6;
    double *dd, *o5;
    size_t f = (size_t) n * 4, g = (size_t) n * 8;
    if (hipMalloc(&da, f) || hipMalloc(&db, f) || hipMalloc(&dd, g) || hipMalloc(&o1, f) || hipMalloc(&o2, f) || hipMalloc(&o3, f) ||
        hipMalloc(&o4, f) || hipMalloc(&o5, g) || hipMalloc(&o6, f))
        return 1;
    (void) hipMemcpy(da, a, f, hipMemcpyHostToDevice);
    (void) hipMemcpy(db, b, f, hipMemcpyHostToDevice);
    (void) hipMemcpy(dd, d, g, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3((n + 255) / 256), dim3(256), 0, 0, n, da, db, dd, o1, o2, o3, o4, o5, o6);
    if (hipDeviceSynchronize() != hipSuccess)
        return 2;
    (void) hipMemcpy(o_atan2, o1, f, hipMemcpyDeviceToHost);
    (void) hipMemcpy(o_asin, o2, f, hipMemcpyDeviceToHost);
    (void) hipMemcpy(o_div, o3, f, hipMemcpyDeviceToHost);
    (void) hipMemcpy(o_sqrtf, o4, f, hipMemcpyDeviceToHost);
    (void) hipMemcpy(o_sqrtd, o5, g, hipMemcpyDeviceToHost);
    (void) hipMemcpy(o_len, o6, f, hipMemcpyDeviceToHost);
    for (void* q : {(void*) da, (void*) db, (void*) dd, (void*) o1, (void*) o2, (void*) o3, (void*) o4, (void*) o5, (void*) o6})
        (void) hipFree(q);
    return 0;
}
