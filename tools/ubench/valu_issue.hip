// Vector-issue roof of the MI355X (gfx950), measured: wave-instructions per clock and SIMD with ALL compute units busy and 1 / 2 / 4 / 8
// wavefronts per SIMD. (tools/ubench/lone_wave.hip times ONE wavefront on one SIMD: an issue cadence, not a throughput. bench.py's
// roofline.valu peak comes from this file's table: profiles/r05_valu_issue.txt.)
//
// Build: hipcc --offload-arch=gfx950 -O2 valu_issue.hip -o valu_issue ; run: ./valu_issue
// Every kernel is a block of 256 threads (one wavefront per SIMD of a CU); the number of blocks a CU holds — hence the wavefronts per SIMD — is
// set by the dynamic LDS a block asks for (160 KiB per CU / w), the grid is 256 CUs x w blocks, so the whole launch is resident at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP2(x) x x
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

constexpr int ITERS = 4000;

// body = 64 instructions per iteration (8 independent chains x 8) unless stated otherwise
#define KERNEL(name, body, clobbers...)                                                                        \
    __global__ __launch_bounds__(256) void name(uint64_t* out, int dummy)                                      \
    {                                                                                                          \
        extern __shared__ int lds[];                                                                           \
        lds[threadIdx.x] = threadIdx.x;                                                                        \
        __syncthreads();                                                                                       \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                      \
        for (int it = 0; it < ITERS; it++)                                                                     \
        {                                                                                                      \
            asm volatile(body ::"v"(threadIdx.x * 4), "s"(dummy) : "memory", clobbers);                        \
        }                                                                                                      \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                      \
        if ((threadIdx.x & 63) == 0)                                                                           \
            out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                                \
    }

#define V8(op, tail) \
    op " v10, v10" tail "\n" op " v11, v11" tail "\n" op " v12, v12" tail "\n" op " v13, v13" tail "\n" \
    op " v14, v14" tail "\n" op " v15, v15" tail "\n" op " v16, v16" tail "\n" op " v17, v17" tail "\n"
#define V8_64(op, tail) \
    op " v[10:11], v[10:11]" tail "\n" op " v[12:13], v[12:13]" tail "\n" op " v[14:15], v[14:15]" tail "\n" op " v[16:17], v[16:17]" tail "\n" \
    op " v[18:19], v[18:19]" tail "\n" op " v[20:21], v[20:21]" tail "\n" op " v[22:23], v[22:23]" tail "\n" op " v[24:25], v[24:25]" tail "\n"
#define CL32 "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v30", "v31"
#define CL64 "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v30", "v31", "v32", "v33"

KERNEL(k_add_u32, REP8(V8("v_add_u32", ", v30")), CL32)
KERNEL(k_add_u32_dep, REP16(REP4("v_add_u32 v10, v10, v30\n")), CL32)
KERNEL(k_mul_f32, REP8(V8("v_mul_f32", ", v30")), CL32)
KERNEL(k_fma_f32, REP8(V8("v_fma_f32", ", v30, v31")), CL32)
KERNEL(k_cndmask, REP8(V8("v_cndmask_b32", ", v30, vcc")), CL32)
KERNEL(k_cndmask_vcc1, "s_mov_b64 vcc, -1\n" REP8(V8("v_cndmask_b32", ", v30, vcc")), CL32, "vcc")
KERNEL(k_cndmask_vcc0, "s_mov_b64 vcc, 0\n" REP8(V8("v_cndmask_b32", ", v30, vcc")), CL32, "vcc")
KERNEL(k_cndmask_vccx, "s_mov_b64 vcc, 0x5555aaaa\n" REP8(V8("v_cndmask_b32", ", v30, vcc")), CL32, "vcc")
KERNEL(k_cndmask_sgpr, "s_mov_b64 s[20:21], 0x5555aaaa\n" REP8(V8("v_cndmask_b32_e64", ", v30, s[20:21]")), CL32, "s20", "s21")
KERNEL(k_cndmask_other_dst, "s_mov_b64 vcc, 0x5555aaaa\n" REP8("v_cndmask_b32 v10, v20, v30, vcc\n v_cndmask_b32 v11, v21, v30, vcc\n v_cndmask_b32 v12, v22, v30, vcc\n v_cndmask_b32 v13, v23, v30, vcc\n v_cndmask_b32 v14, v20, v30, vcc\n v_cndmask_b32 v15, v21, v30, vcc\n v_cndmask_b32 v16, v22, v30, vcc\n v_cndmask_b32 v17, v23, v30, vcc\n"), CL32, "vcc", "v20", "v21", "v22", "v23")
KERNEL(k_cmp_cndmask, REP8(REP4("v_cmp_gt_f32 vcc, v10, v30\n v_cndmask_b32 v11, v11, v30, vcc\n")), CL32, "vcc")
KERNEL(k_cmp_sgpr_cndmask, REP8("v_cmp_gt_f32_e64 s[20:21], v10, v30\n v_cmp_gt_f32_e64 s[22:23], v11, v30\n v_cmp_gt_f32_e64 s[24:25], v12, v30\n v_cmp_gt_f32_e64 s[26:27], v13, v30\n v_cndmask_b32_e64 v14, v14, v30, s[20:21]\n v_cndmask_b32_e64 v15, v15, v30, s[22:23]\n v_cndmask_b32_e64 v16, v16, v30, s[24:25]\n v_cndmask_b32_e64 v17, v17, v30, s[26:27]\n"), CL32, "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
KERNEL(k_add2_cnd2, "s_mov_b64 vcc, 0x5555aaaa\n" REP8(REP2("v_add_u32 v10, v10, v30\n v_add_u32 v12, v12, v30\n v_cndmask_b32 v11, v11, v30, vcc\n v_cndmask_b32 v13, v13, v30, vcc\n")), CL32, "vcc")
KERNEL(k_add6_cnd2, "s_mov_b64 vcc, 0x5555aaaa\n" REP8("v_add_u32 v10, v10, v30\n v_add_u32 v12, v12, v30\n v_add_u32 v14, v14, v30\n v_add_u32 v15, v15, v30\n v_add_u32 v16, v16, v30\n v_add_u32 v17, v17, v30\n v_cndmask_b32 v11, v11, v30, vcc\n v_cndmask_b32 v13, v13, v30, vcc\n"), CL32, "vcc")
KERNEL(k_cmp_cnd2, REP8(REP2("v_cmp_gt_f32 vcc, v10, v30\n v_cndmask_b32 v11, v11, v30, vcc\n v_cndmask_b32 v13, v13, v30, vcc\n v_add_u32 v12, v12, v30\n")), CL32, "vcc")
KERNEL(k_add_cndmask, "s_mov_b64 vcc, 0x5555aaaa\n" REP8(REP4("v_add_u32 v10, v10, v30\n v_cndmask_b32 v11, v11, v30, vcc\n")), CL32, "vcc")
KERNEL(k_max_f32, REP8(V8("v_max_f32", ", v30")), CL32)
KERNEL(k_min_u32, REP8(V8("v_min_u32", ", v30")), CL32)
KERNEL(k_add_f32, REP8(V8("v_add_f32", ", v30")), CL32)
KERNEL(k_add_u32_sgpr, REP8(V8("v_add_u32", ", s24")), CL32, "s24")
KERNEL(k_add_co, REP8("v_add_co_u32 v10, vcc, v10, v30\n v_add_co_u32 v11, vcc, v11, v30\n v_add_co_u32 v12, vcc, v12, v30\n v_add_co_u32 v13, vcc, v13, v30\n v_add_co_u32 v14, vcc, v14, v30\n v_add_co_u32 v15, vcc, v15, v30\n v_add_co_u32 v16, vcc, v16, v30\n v_add_co_u32 v17, vcc, v17, v30\n"), CL32, "vcc")
KERNEL(k_bfe, REP8(V8("v_bfe_u32", ", 3, 5")), CL32)
KERNEL(k_sqrt_f32, REP8("v_sqrt_f32 v10, v10\n v_sqrt_f32 v11, v11\n v_sqrt_f32 v12, v12\n v_sqrt_f32 v13, v13\n v_sqrt_f32 v14, v14\n v_sqrt_f32 v15, v15\n v_sqrt_f32 v16, v16\n v_sqrt_f32 v17, v17\n"), CL32)
KERNEL(k_rcp_f64, REP8("v_rcp_f64 v[10:11], v[10:11]\n v_rcp_f64 v[12:13], v[12:13]\n v_rcp_f64 v[14:15], v[14:15]\n v_rcp_f64 v[16:17], v[16:17]\n v_rcp_f64 v[18:19], v[18:19]\n v_rcp_f64 v[20:21], v[20:21]\n v_rcp_f64 v[22:23], v[22:23]\n v_rcp_f64 v[24:25], v[24:25]\n"), CL64)
KERNEL(k_cvt_f64_f32, REP8("v_cvt_f64_f32 v[10:11], v30\n v_cvt_f64_f32 v[12:13], v30\n v_cvt_f64_f32 v[14:15], v30\n v_cvt_f64_f32 v[16:17], v30\n v_cvt_f64_f32 v[18:19], v30\n v_cvt_f64_f32 v[20:21], v30\n v_cvt_f64_f32 v[22:23], v30\n v_cvt_f64_f32 v[24:25], v30\n"), CL64)
KERNEL(k_and_b32, REP8(V8("v_and_b32", ", v30")), CL32)
KERNEL(k_lshl_add_u32, REP8(V8("v_lshl_add_u32", ", 1, v30")), CL32)
KERNEL(k_cmp_f32, REP16(REP4("v_cmp_gt_f32 vcc, v10, v30\n")), CL32, "vcc")
KERNEL(k_mul_lo_u32, REP8(V8("v_mul_lo_u32", ", v30")), CL32)
KERNEL(k_rcp_f32, REP8("v_rcp_f32 v10, v10\n v_rcp_f32 v11, v11\n v_rcp_f32 v12, v12\n v_rcp_f32 v13, v13\n v_rcp_f32 v14, v14\n v_rcp_f32 v15, v15\n v_rcp_f32 v16, v16\n v_rcp_f32 v17, v17\n"), CL32)
KERNEL(k_add_f64, REP8(V8_64("v_add_f64", ", v[30:31]")), CL64)
KERNEL(k_mul_f64, REP8(V8_64("v_mul_f64", ", v[30:31]")), CL64)
KERNEL(k_fma_f64, REP8(V8_64("v_fma_f64", ", v[30:31], v[32:33]")), CL64)
KERNEL(k_lshl_add_u64, REP8(V8_64("v_lshl_add_u64", ", 0, v[30:31]")), CL64)
KERNEL(k_pk_mul_f32, REP8(V8_64("v_pk_mul_f32", ", v[30:31]")), CL64)
KERNEL(k_pk_add_f32, REP8(V8_64("v_pk_add_f32", ", v[30:31]")), CL64)
KERNEL(k_pk_fma_f32, REP8(V8_64("v_pk_fma_f32", ", v[30:31], v[32:33]")), CL64)
KERNEL(k_mov_dpp, REP8(V8("v_mov_b32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf")), CL32)
KERNEL(k_readlane, REP16(REP4("v_readlane_b32 s20, v10, 3\n")), CL32, "s20")
KERNEL(k_salu, REP16("s_add_u32 s20, s20, s24\n s_add_u32 s21, s21, s24\n s_add_u32 s22, s22, s24\n s_add_u32 s23, s23, s24\n"), "s20", "s21", "s22", "s23", "s24", "scc")
// 4 VALU : 2 SALU : 1 ds_read per 7 instructions, 8 times + one wait = 57 instructions (56 counted: 32 VALU)
KERNEL(k_mix, "v_mov_b32 v20, %0\n" REP8("v_add_u32 v10, v10, v30\n s_add_u32 s20, s20, s24\n v_mul_f32 v11, v11, v30\n ds_read_b32 v21, v20\n v_cndmask_b32 v12, v12, v30, vcc\n s_and_b32 s21, s21, s24\n v_add_u32 v13, v13, v30\n") "s_waitcnt lgkmcnt(0)\n",
       CL32, "v20", "v21", "s20", "s21", "s24", "scc")
// the shape of the path's kernels: VALU with a dependent chain of length 2 and a branch every 16
KERNEL(k_dep2, REP4(REP4("v_add_u32 v10, v10, v30\n v_add_u32 v10, v10, v30\n v_add_u32 v11, v11, v30\n v_add_u32 v11, v11, v30\n")), CL32)

struct Case
{
    const char* name;
    void (*fn)(uint64_t*, int);
    int per_iter; // instructions counted per iteration
    const char* note;
};

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("# %s: %d CUs, %d SIMDs, reported shader clock %.0f MHz, LDS per CU %zu KB\n", prop.gcnArchName, cus, cus * 4, clk_khz / 1000., (size_t) prop.maxSharedMemoryPerMultiProcessor / 1024);
    uint64_t* out;
    hipMalloc(&out, 8 * 4 * cus * 8 + 64);
    std::vector<Case> cases = {
        {"v_add_u32 (8 independent chains)", k_add_u32, 64, ""},
        {"v_add_u32 (ONE dependent chain)", k_add_u32_dep, 64, ""},
        {"v_add_u32 (2 chains, pairs dependent)", k_dep2, 64, ""},
        {"v_mul_f32", k_mul_f32, 64, ""},
        {"v_fma_f32", k_fma_f32, 64, ""},
        {"v_cndmask_b32", k_cndmask, 64, ""},
        {"v_cndmask_b32 vcc = -1", k_cndmask_vcc1, 64, ""},
        {"v_cndmask_b32 vcc = 0", k_cndmask_vcc0, 64, ""},
        {"v_cndmask_b32 vcc = pattern", k_cndmask_vccx, 64, ""},
        {"v_cndmask_b32_e64 mask in s[20:21]", k_cndmask_sgpr, 64, ""},
        {"v_cndmask_b32 dst != src", k_cndmask_other_dst, 64, ""},
        {"v_cmp_gt_f32 vcc ; v_cndmask vcc (pairs)", k_cmp_cndmask, 64, ""},
        {"4 x v_cmp -> sgpr pairs ; 4 x v_cndmask_e64", k_cmp_sgpr_cndmask, 64, ""},
        {"v_add_u32 ; v_cndmask vcc (alternating)", k_add_cndmask, 64, ""},
        {"2 x v_add_u32 ; 2 x v_cndmask vcc", k_add2_cnd2, 64, ""},
        {"6 x v_add_u32 ; 2 x v_cndmask vcc", k_add6_cnd2, 64, ""},
        {"v_cmp vcc ; 2 x v_cndmask vcc ; v_add_u32", k_cmp_cnd2, 64, ""},
        {"v_max_f32", k_max_f32, 64, ""},
        {"v_min_u32", k_min_u32, 64, ""},
        {"v_add_f32", k_add_f32, 64, ""},
        {"v_add_u32 v, v, sgpr", k_add_u32_sgpr, 64, ""},
        {"v_add_co_u32 (writes vcc)", k_add_co, 64, ""},
        {"v_bfe_u32 (VOP3, inline constants)", k_bfe, 64, ""},
        {"v_sqrt_f32", k_sqrt_f32, 64, ""},
        {"v_rcp_f64", k_rcp_f64, 64, ""},
        {"v_cvt_f64_f32", k_cvt_f64_f32, 64, ""},
        {"v_and_b32", k_and_b32, 64, ""},
        {"v_lshl_add_u32", k_lshl_add_u32, 64, ""},
        {"v_cmp_gt_f32 -> vcc", k_cmp_f32, 64, ""},
        {"v_mul_lo_u32", k_mul_lo_u32, 64, ""},
        {"v_rcp_f32", k_rcp_f32, 64, ""},
        {"v_add_f64", k_add_f64, 64, ""},
        {"v_mul_f64", k_mul_f64, 64, ""},
        {"v_fma_f64", k_fma_f64, 64, ""},
        {"v_lshl_add_u64", k_lshl_add_u64, 64, ""},
        {"v_pk_mul_f32 (2 f32 per lane)", k_pk_mul_f32, 64, ""},
        {"v_pk_add_f32", k_pk_add_f32, 64, ""},
        {"v_pk_fma_f32", k_pk_fma_f32, 64, ""},
        {"v_mov_b32_dpp row_shr:1", k_mov_dpp, 64, ""},
        {"v_readlane_b32", k_readlane, 64, ""},
        {"s_add_u32 (4 chains)", k_salu, 64, ""},
        {"mix 4 VALU : 2 SALU : 1 ds_read (all 56 counted)", k_mix, 56, ""},
    };
    const int lds_total = 160 * 1024;
    printf("%-52s %s\n", "instruction", "wave-instructions per clock (reported shader clock) and SIMD at 1 / 2 / 4 / 8 wavefronts per SIMD, by wall time   [s_memtime ticks per wave-instruction of the slowest wavefront at 1 / 8]   {G wave-instr/s of the whole GPU at 8}");
    for (auto& c : cases)
    {
        double rate[4], percy[4], gips[4];
        int wi = 0;
        for (int w : {1, 2, 4, 8})
        {
            int lds = lds_total / w - 2048; // one block's share: w blocks fit a CU, w + 1 do not
            if (hipFuncSetAttribute((const void*) c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            {
                (void) hipGetLastError();
                lds = std::min(lds, 64 * 1024); // (then nothing but the even spread of cus * w blocks keeps a CU at w blocks)
            }
            const int grid = cus * w;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipFuncSetAttribute((const void*) c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            float best_ms = 1e9f;
            uint64_t best_ticks = 0;
            for (int rep = 0; rep < 3; rep++)
            {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(c.fn, dim3(grid), dim3(256), lds, 0, out, 1);
                hipEventRecord(e1, 0);
                if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess)
                {
                    printf("launch failed: %s w %d lds %d\n", c.name, w, lds);
                    break;
                }
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                std::vector<uint64_t> h((size_t) grid * 4);
                hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
                uint64_t mx = 0;
                for (auto v : h)
                    mx = std::max(mx, v);
                if (ms < best_ms)
                {
                    best_ms = ms;
                    best_ticks = mx;
                }
            }
            const double n_wave = (double) c.per_iter * ITERS;          // wave-instructions per wavefront
            rate[wi] = n_wave * w / (best_ms * 1e-3 * clk_khz * 1e3);  // per SIMD and clock of the reported shader clock, by wall time (launch included)
            percy[wi] = (double) best_ticks / n_wave;
            gips[wi] = n_wave * w * cus * 4 / (best_ms * 1e-3) / 1e9;
            wi++;
        }
        printf("%-52s %6.3f %6.3f %6.3f %6.3f   [%5.2f %5.2f]   {%7.1f}\n", c.name, rate[0], rate[1], rate[2], rate[3], percy[0], percy[3], gips[3]);
    }
    return 0;
}
