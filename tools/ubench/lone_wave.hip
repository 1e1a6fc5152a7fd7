// Lone-wave instruction cost model for gfx950: cycles per instruction pattern when ONE wavefront runs on a SIMD (the situation of
// the serial per-stream kernels k_insert2 / k_seg_scan / k_assoc_lds). Build: hipcc --offload-arch=gfx950 -O2 lone_wave.hip -o lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

constexpr int ITERS = 2000;

#define KERNEL(name, n_per_iter, body, clobbers...)                                           \
    __global__ void name(uint64_t* out, int* buf, int dummy)                                  \
    {                                                                                         \
        __shared__ int lds[1024];                                                             \
        for (int i = threadIdx.x; i < 1024; i += 64)                                          \
            lds[i] = ((i + 7) & 255) * 4;                                                     \
        __syncthreads();                                                                      \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                           \
        for (int it = 0; it < ITERS; it++)                                                    \
        {                                                                                     \
            asm volatile(body ::"v"(threadIdx.x * 4), "s"(dummy), "v"(buf) : "memory", clobbers); \
        }                                                                                     \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                           \
        if (threadIdx.x == 0)                                                                 \
        {                                                                                     \
            out[0] = t1 - t0;                                                                 \
            out[1] = (uint64_t) n_per_iter * ITERS;                                           \
        }                                                                                     \
    }

KERNEL(k_valu_dep, 64, REP64("v_add_u32 v10, v10, v11\n"), "v10", "v11")
KERNEL(k_valu_indep, 64, REP16("v_add_u32 v10, v10, v20\n v_add_u32 v11, v11, v20\n v_add_u32 v12, v12, v20\n v_add_u32 v13, v13, v20\n"), "v10", "v11", "v12", "v13", "v20")
KERNEL(k_valu64_dep, 64, REP64("v_lshl_add_u64 v[10:11], v[10:11], 0, v[12:13]\n"), "v10", "v11", "v12", "v13")
KERNEL(k_valu_f64_dep, 64, REP64("v_add_f64 v[10:11], v[10:11], v[12:13]\n"), "v10", "v11", "v12", "v13")
KERNEL(k_salu_dep, 64, REP64("s_add_u32 s20, s20, s21\n"), "s20", "s21", "scc")
KERNEL(k_salu_indep, 64, REP16("s_add_u32 s20, s20, s24\n s_add_u32 s21, s21, s24\n s_add_u32 s22, s22, s24\n s_add_u32 s23, s23, s24\n"), "s20", "s21", "s22", "s23", "s24", "scc")
KERNEL(k_mix_valu_salu, 64, REP16("v_add_u32 v10, v10, v11\n s_add_u32 s20, s20, s21\n v_add_u32 v10, v10, v11\n s_add_u32 s20, s20, s21\n"), "v10", "v11", "s20", "s21", "scc")
KERNEL(k_readfirstlane_chain, 48, REP16("v_readfirstlane_b32 s20, v10\n s_add_u32 s20, s20, 1\n v_mov_b32 v10, s20\n"), "v10", "s20", "scc")
KERNEL(k_vcmp_branch_nottaken, 32, REP16("v_cmp_eq_u32 vcc, v10, v10\n s_cbranch_vccz 1f\n") "1:\n", "v10", "vcc")
KERNEL(k_scmp_branch_nottaken, 32, REP16("s_cmp_eq_u32 s20, s20\n s_cbranch_scc0 1f\n") "1:\n", "s20", "scc")
KERNEL(k_scmp_branch_taken, 32, REP16("s_cmp_eq_u32 s20, s20\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n"), "s20", "scc")
KERNEL(k_branch_taken_far, 16,
       REP16("s_cmp_eq_u32 s20, s20\n s_cbranch_scc1 1f\n" REP16("s_nop 0\n") "1:\n"), "s20", "scc")
KERNEL(k_saveexec_skip, 48, REP16("v_cmp_ne_u32 vcc, v10, v10\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 v11, v11, v11\n 1:\n s_or_b64 exec, exec, s[20:21]\n"),
       "v10", "v11", "s20", "s21", "vcc")
KERNEL(k_saveexec_noskip, 64, REP16("v_cmp_eq_u32 vcc, v10, v10\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 v11, v11, v11\n 1:\n s_or_b64 exec, exec, s[20:21]\n"),
       "v10", "v11", "s20", "s21", "vcc")
KERNEL(k_ds_read_chain, 16, "v_mov_b32 v10, %0\n" REP16("ds_read_b32 v10, v10\n s_waitcnt lgkmcnt(0)\n"), "v10")
KERNEL(k_ds_read_4indep, 64, "v_mov_b32 v10, %0\n" REP16("ds_read_b32 v11, v10\n ds_read_b32 v12, v10 offset:256\n ds_read_b32 v13, v10 offset:512\n ds_read_b32 v14, v10 offset:768\n s_waitcnt lgkmcnt(0)\n"),
       "v10", "v11", "v12", "v13", "v14")
KERNEL(k_ds_bpermute_chain, 16, "v_mov_b32 v10, %0\n" REP16("ds_bpermute_b32 v10, v10, v10\n s_waitcnt lgkmcnt(0)\n"), "v10")
KERNEL(k_ds_write_read, 32, "v_mov_b32 v10, %0\n" REP16("ds_write_b32 v10, v10\n ds_read_b32 v11, v10\n s_waitcnt lgkmcnt(0)\n"), "v10", "v11")
KERNEL(k_ds_write_wait, 16, "v_mov_b32 v10, %0\n" REP16("ds_write_b32 v10, v10\n s_waitcnt lgkmcnt(0)\n"), "v10")
KERNEL(k_ds_atomic_noret_wait, 16, "v_mov_b32 v10, %0\n" REP16("ds_max_u32 v10, v10\n s_waitcnt lgkmcnt(0)\n"), "v10")
KERNEL(k_ds_atomic_same_addr, 16, "v_mov_b32 v10, 0\n" REP16("ds_add_u32 v10, v10\n s_waitcnt lgkmcnt(0)\n"), "v10")
KERNEL(k_ds_read_b64_chain, 16, "v_mov_b32 v10, %0\n" REP16("ds_read_b64 v[10:11], v10\n s_waitcnt lgkmcnt(0)\n"), "v10", "v11")
KERNEL(k_global_load_chain, 16, REP16("global_load_dword v12, %2, off\n s_waitcnt vmcnt(0)\n"), "v12")
KERNEL(k_global_store_wait, 16, "v_mov_b32 v12, %0\n" REP16("global_store_dword %2, v12, off offset:1024\n s_waitcnt vmcnt(0)\n"), "v12")
KERNEL(k_readlane_spill, 32, REP16("v_writelane_b32 v10, s20, 3\n v_readlane_b32 s20, v10, 3\n"), "v10", "s20")
KERNEL(k_ballot_like, 48, REP16("v_cmp_gt_u32 vcc, v10, v11\n s_bcnt1_i32_b64 s20, vcc\n v_add_u32 v10, s20, v10\n"), "v10", "v11", "s20", "vcc", "scc")
KERNEL(k_smemtime, 16, REP16("s_memtime s[20:21]\n s_waitcnt lgkmcnt(0)\n"), "s20", "s21")
KERNEL(k_snop_loop, 64, REP64("s_nop 0\n"), "s20")

struct Case
{
    const char* name;
    void (*fn)(uint64_t*, int*, int);
};

int main()
{
    uint64_t* out;
    int* buf;
    hipMalloc(&out, 64);
    hipMalloc(&buf, 1 << 20);
    hipMemset(buf, 0, 1 << 20);
    std::vector<Case> cases = {
        {"valu dependent (v_add_u32)", k_valu_dep}, {"valu independent x4", k_valu_indep}, {"valu 64-bit add dependent", k_valu64_dep},
        {"valu f64 add dependent", k_valu_f64_dep}, {"salu dependent", k_salu_dep}, {"salu independent x4", k_salu_indep},
        {"valu/salu interleaved (2 chains)", k_mix_valu_salu}, {"readfirstlane->salu->v_mov chain", k_readfirstlane_chain},
        {"v_cmp + s_cbranch_vccz not taken", k_vcmp_branch_nottaken}, {"s_cmp + s_cbranch not taken", k_scmp_branch_nottaken},
        {"s_cmp + s_cbranch taken (skip 1)", k_scmp_branch_taken}, {"s_cmp + s_cbranch taken (skip 16)", k_branch_taken_far},
        {"saveexec + execz skip taken", k_saveexec_skip}, {"saveexec + execz not taken", k_saveexec_noskip},
        {"ds_read_b32 dependent + wait", k_ds_read_chain}, {"4 x ds_read_b32 + one wait", k_ds_read_4indep},
        {"ds_bpermute dependent + wait", k_ds_bpermute_chain}, {"ds_write ; ds_read ; wait", k_ds_write_read},
        {"ds_write + wait", k_ds_write_wait}, {"ds_max_u32 (no return) + wait", k_ds_atomic_noret_wait},
        {"ds_add_u32 64 lanes same address + wait", k_ds_atomic_same_addr}, {"ds_read_b64 dependent + wait", k_ds_read_b64_chain},
        {"global_load dependent (L2 hit) + wait", k_global_load_chain}, {"global_store + wait vmcnt(0)", k_global_store_wait},
        {"v_writelane + v_readlane (sgpr spill)", k_readlane_spill}, {"v_cmp -> s_bcnt1 -> valu", k_ballot_like},
        {"s_memtime + wait", k_smemtime}, {"s_nop 0", k_snop_loop},
    };
    for (auto& c : cases)
    {
        uint64_t h[2];
        for (int rep = 0; rep < 2; rep++)
        {
            hipLaunchKernelGGL(c.fn, dim3(1), dim3(64), 0, 0, out, buf, 1);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("%-44s %8.1f ticks per instruction (%llu ticks / %llu)\n", c.name, (double) h[0] / (double) h[1], (unsigned long long) h[0],
               (unsigned long long) h[1]);
    }
    return 0;
}
