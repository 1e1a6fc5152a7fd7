// does a CU-masked stream confine a kernel? Every block records (XCC_ID, SE, CU) of the wave it runs on; the host counts distinct ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned* out)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < 200000) {}
    if (threadIdx.x == 0)
        out[blockIdx.x] = (xcc & 0xf) << 16 | ((hw >> 13) & 0x7) << 8 | ((hw >> 8) & 0xf);
}
int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    printf("CUs %d\n", ncu);
    unsigned* d;
    const int nb = 4096;
    hipMalloc(&d, nb * 4);
    for (int cnt : {256, 128, 64, 16})
    {
        std::vector<uint32_t> m(words, 0u);
        for (int i = 0; i < cnt && i < ncu; i++)
            m[i >> 5] |= 1u << (i & 31);
        hipStream_t s;
        hipError_t rc = hipExtStreamCreateWithCUMask(&s, words, m.data());
        hipEvent_t a, b;
        hipEventCreate(&a), hipEventCreate(&b);
        hipEventRecord(a, s);
        where<<<nb, 64, 0, s>>>(d);
        hipEventRecord(b, s);
        hipStreamSynchronize(s);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned> h(nb);
        hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> cus, xccs;
        for (unsigned v : h)
            cus.insert(v), xccs.insert(v >> 16);
        printf("mask %3d CUs: rc %d, distinct (xcc,se,cu) %zu, xccs %zu, %.3f ms\n", cnt, (int) rc, cus.size(), xccs.size(), ms);
        hipStreamDestroy(s);
    }
    return 0;
}
