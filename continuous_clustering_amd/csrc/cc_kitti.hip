// cc_kitti.hip — KITTI frame -> pseudo-firings on gfx950 (include/cc_kitti.h).
//
// Per frame (one .bin cloud, n ~ 120 k points, 16 B each) three kernels:
//
//   k_kitti_points   one 1024-thread block per frame, one pass over the points in file order:
//                      * atan2f of the point (bit-exact glibc float atan2f, cc_math.h — the reference's std::atan2(float,float)),
//                      * recoverLaserIndices (kitti_loader.cpp:48-99): "row += 1 at every azimuth jump" is a prefix sum of jump
//                        flags; the block scans 1024 points per step and carries (last azimuth, row) to the next step,
//                      * undoEgoMotionCorrection (:199-209): bin lookup + 3x4 double transform, evaluated in Eigen's order,
//                      * the column index of generateRangeImage (:118-126) from the un-corrected point,
//                      * a 64-bin row histogram for the next kernel.
//   k_kitti_image    one wavefront per frame. generateRangeImage's shift-if-occupied rule (:129-160) depends on the insertion
//                    order inside a row and on nothing else, so lane r replays row r in file order: a stable counting sort by
//                    row (wave ballots) gives every lane its run of points, an LDS bitmap (64 rows x 2200 bits) is the occupancy,
//                    and the winner of a cell is simply the last store to it.
//   k_kitti_firings  one thread per cell, laser index fastest: gathers the winner and writes the firing arrays in the layout of
//                    cc_engine_add_firings_device (makePseudoFiringFromRangeImageColumn, kitti_demo.cpp:123-159).
//
// Algorithmic HBM bytes per frame: 16 n (read .bin) + 16 n + 3 n (un-corrected points, row, column) + 2*(4+2) n (sort) + 64*2200*
// (4 + 4 + 12 + 1 + 4) = ~50 n + 3.5 MB, i.e. ~9.5 MB for n = 120 k.
//
// The host functions at the bottom are the pose arithmetic of the same call sites (Eigen::Isometry3d / Quaterniond of Eigen 3.4,
// restated: Eigen is not a dependency). Device code is built with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cc_hip.h"
#include "../../include/cc_kitti.h"
#include "cc_math.h"

namespace
{

thread_local std::string g_kitti_error;

int fail(int code, const std::string& what)
{
    g_kitti_error = what;
    return code;
}

#define KITTI_HIP_CHECK(expr)                                                                          \
    do                                                                                                 \
    {                                                                                                  \
        hipError_t err__ = (expr);                                                                     \
        if (err__ != hipSuccess)                                                                       \
            return fail(CC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__));             \
    } while (0)

constexpr int ROWS = CC_KITTI_ROWS, COLS = CC_KITTI_COLS;
constexpr int OCC_WORDS = 69; // ceil(2200 / 32), odd so that the 64 lanes' rows start in different LDS banks
constexpr unsigned short COL_SKIP = 0xFFFF;

// device-side description of one frame slot
struct FrameSlot
{
    const float4* pts_in;
    float4* pts;          // after undoEgoMotionCorrection
    unsigned char* laser; // row per point
    unsigned short* col;  // generateRangeImage column per point (COL_SKIP: NaN azimuth)
    int* order;           // point indices, stably sorted by row
    unsigned short* ocol; // their columns, same order
    int* cell_src;        // [2200][64] winner per cell, -1 = empty
    int* hist;            // [64] points per row (the tail after the 65th row counted in row 0)
    long long* info;      // [4] jumps, break index, max_columns, skipped
    const double* bins;   // [num_bins][12]
    long long n;
    double duration;
    int num_bins;
    unsigned stages;
    float* o_xyz;
    unsigned char* o_int;
    int* o_orig;
};

__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d)
            v += o;
    }
    return v;
}

__global__ __launch_bounds__(1024) void k_kitti_points(const FrameSlot* slots)
{
    const FrameSlot F = slots[blockIdx.x];
    __shared__ double s_mono[1025];
    __shared__ int s_wsum[16];
    __shared__ int s_hist[ROWS];
    __shared__ int s_rowcnt[ROWS]; // points per row before the break (the max_columns statistic)
    __shared__ long long s_break;
    __shared__ unsigned long long s_skipped;
    __shared__ int s_carry_row, s_jumps;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid < ROWS)
    {
        s_hist[tid] = 0;
        s_rowcnt[tid] = 0;
    }
    if (tid == 0)
    {
        s_mono[0] = -1.0; // prev_azimuth_monotonic = -1 (kitti_loader.cpp:56)
        s_break = F.n;
        s_skipped = 0;
        s_carry_row = 0;
        s_jumps = 0;
    }
    __syncthreads();
    const bool recover = F.stages & CC_KITTI_RECOVER_ROWS, undo = (F.stages & CC_KITTI_UNDO_EGO_MOTION) && F.num_bins > 0;
    const double two_pi = 2 * M_PI;
    const double column_width = (2 * M_PI) / COLS; // kitti_loader.cpp:105
    for (long long base = 0; base < F.n; base += 1024)
    {
        const long long i = base + tid;
        const bool live = i < F.n;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        double az = 0.0;
        if (live)
        {
            p = F.pts_in[i];
            az = (double) ccm::atan2f_exact(p.y, p.x); // std::atan2(float, float) -> float -> double (:63, :199)
        }
        int row = 0;
        if (recover)
        {
            // 0 -> pi -> -pi -> 0 made monotonic 0 -> 2 pi (:67)
            const double mono = az < 0 ? az + two_pi : az;
            if (live)
                s_mono[tid + 1] = mono;
            __syncthreads();
            const double prev = s_mono[tid];
            const int jump = live && prev >= 0 && mono - prev < -0.7; // :72
            // inclusive scan of the jump flags over the block
            const int incl = wave_incl_scan(jump);
            if (lane == 63)
                s_wsum[wid] = incl;
            __syncthreads();
            int before = s_carry_row;
            for (int w = 0; w < wid; w++)
                before += s_wsum[w];
            const int raw = before + incl;
            __syncthreads();
            if (tid == 1023)
            {
                s_carry_row = raw;
                s_mono[0] = mono; // live for every full step; the last, partial step has no successor
            }
            if (live)
            {
                if (raw >= ROWS) // :75-76: the loop ends, this point and the rest keep laser_index 0
                {
                    atomicMin((unsigned long long*) &s_break, (unsigned long long) i);
                    row = 0;
                }
                else
                {
                    row = raw;
                    atomicAdd(&s_rowcnt[row], 1);
                }
            }
            if (live && i == F.n - 1)
                s_jumps = raw; // jumps seen if the loop had not stopped
        }
        else if (live)
            row = F.laser ? F.laser[i] & (ROWS - 1) : 0;

        if (live)
        {
            if (undo)
            {
                // :199-208
                const double fraction = (M_PI - az) / (2.0 * M_PI);
                const double b = (fraction * F.duration) / 1000000.0;
                int bin = b >= 0 ? (b < (double) F.num_bins ? (int) b : F.num_bins - 1) : 0; // NaN and the exact-end case clamp
                const double* T = F.bins + (size_t) bin * 12;
                const double x = p.x, y = p.y, z = p.z;
                const double ux = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
                const double uy = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
                const double uz = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
                p.x = (float) ux;
                p.y = (float) uy;
                p.z = (float) uz;
            }
            unsigned short c = COL_SKIP;
            const float azf = undo ? ccm::atan2f_exact(p.y, p.x) : (float) az;
            if (azf == azf)
            {
                int ci = (int) ((M_PI - (double) azf) / column_width); // :121
                if (ci == COLS)                                          // :124-125
                    ci--;
                c = (unsigned short) ci;
            }
            else
                atomicAdd(&s_skipped, 1ull);
            F.pts[i] = p;
            F.col[i] = c;
            if (recover)
                F.laser[i] = (unsigned char) row;
            atomicAdd(&s_hist[row], 1);
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid < ROWS)
        F.hist[tid] = s_hist[tid];
    if (tid == 0)
    {
        long long jumps = 0;
        int max_columns = 0;
        if (recover && F.n > 0)
        {
            jumps = s_jumps;
            // completed rows feed the statistic at the jump that ends them (:79-80); the jump that breaks does not (:74-76)
            const int completed = (int) (jumps < ROWS - 1 ? jumps : ROWS - 1);
            for (int r = 0; r < completed; r++)
                max_columns = max(max_columns, s_rowcnt[r]);
        }
        F.info[0] = jumps;
        F.info[1] = s_break;
        F.info[2] = max_columns;
        F.info[3] = (long long) s_skipped;
    }
}

__device__ __forceinline__ int ld_agent_i32(const int* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned short ld_agent_u16(const unsigned short* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64) void k_kitti_image(const FrameSlot* slots)
{
    const FrameSlot F = slots[blockIdx.x];
    if (!(F.stages & CC_KITTI_RANGE_IMAGE))
        return;
    __shared__ unsigned s_occ[ROWS * OCC_WORDS];
    __shared__ int s_cur[ROWS];
    const int lane = threadIdx.x;
    for (int w = lane; w < ROWS * OCC_WORDS; w += 64)
        s_occ[w] = 0;
    // run of every row in the sorted order
    const int cnt = F.hist[lane];
    const int start = wave_incl_scan(cnt) - cnt;
    s_cur[lane] = start;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();

    // ---- stable counting sort by row: ranks inside a group of 64 points come from ballots --------------------------------
    for (long long base = 0; base < F.n; base += 64)
    {
        const long long i = base + lane;
        const bool live = i < F.n;
        const int key = live ? F.laser[i] : -1;
        const unsigned short c = live ? F.col[i] : 0;
        unsigned long long todo = __ballot(live);
        while (todo)
        {
            const int leader = __ffsll((long long) todo) - 1;
            const int k = __shfl(key, leader, 64);
            const unsigned long long m = __ballot(key == k) & todo;
            const int first = s_cur[k];
            if (key == k && live)
            {
                const int pos = first + __popcll(m & ((1ull << lane) - 1ull));
                F.order[pos] = (int) i;
                F.ocol[pos] = c;
            }
            if (lane == leader) // LDS accesses of one wavefront execute in program order: the next round reads the new value
                s_cur[k] = first + __popcll(m);
            todo &= ~m;
        }
    }
    __threadfence();
    __builtin_amdgcn_s_barrier();

    // ---- lane r inserts row r in file order (kitti_loader.cpp:115-165) -----------------------------------------------------
    const bool shift = F.stages & CC_KITTI_SHIFT_OCCUPIED;
    unsigned* occ = s_occ + lane * OCC_WORDS;
    int longest = cnt;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        longest = max(longest, __shfl_xor(longest, d, 64));
    constexpr int B = 4;
    for (int s0 = 0; s0 < longest; s0 += B)
    {
        int idx[B];
        unsigned short cc[B];
#pragma unroll
        for (int j = 0; j < B; j++)
        {
            const bool on = s0 + j < cnt;
            idx[j] = on ? ld_agent_i32(F.order + start + s0 + j) : -1;
            cc[j] = on ? ld_agent_u16(F.ocol + start + s0 + j) : COL_SKIP;
        }
#pragma unroll
        for (int j = 0; j < B; j++)
        {
            int c = cc[j];
            if (c == COL_SKIP)
                continue;
            if (shift)
            {
                const int r = c + 1 < COLS ? c + 1 : c, l = c > 0 ? c - 1 : c;
                const unsigned wc = occ[c >> 5], wr = occ[r >> 5], wl = occ[l >> 5];
                const bool occupied = (wc >> (c & 31)) & 1u;
                const bool right_free = c + 1 < COLS && !((wr >> (r & 31)) & 1u);
                const bool left_free = c > 0 && !((wl >> (l & 31)) & 1u);
                if (occupied)
                    c = right_free ? r : (left_free ? l : c); // :137-158
            }
            atomicOr(&occ[c >> 5], 1u << (c & 31));
            F.cell_src[c * ROWS + lane] = idx[j]; // original_kitti_index of the point now in the cell (:165-167)
        }
    }
}

__global__ __launch_bounds__(256) void k_kitti_firings(const FrameSlot* slots)
{
    const FrameSlot F = slots[blockIdx.y];
    if (!(F.stages & CC_KITTI_FIRINGS))
        return;
    const int t = blockIdx.x * 256 + threadIdx.x; // column * 64 + laser
    if (t >= ROWS * COLS)
        return;
    const int src = F.cell_src[t];
    float x, y, z;
    unsigned char inten = 0;
    if (src >= 0)
    {
        const float4 p = F.pts[src];
        x = p.x;
        y = p.y;
        z = p.z;
        // static_cast<uint8_t>(kitti_point.i * 255) (kitti_demo.cpp:148): x86-64 converts through a 32-bit integer and keeps
        // the low byte; values outside the int32 range (and NaN) give 0x80000000 there, low byte 0
        const float v = p.w * 255;
        const int iv = (v > -2147483648.f && v < 2147483648.f) ? (int) v : (int) 0x80000000;
        inten = (unsigned char) (iv & 0xff);
    }
    else
    {
        x = y = z = __builtin_nanf(""); // KittiPoint default (kitti_loader.hpp:30-33)
    }
    if (F.o_xyz)
    {
        F.o_xyz[(size_t) t * 3 + 0] = x;
        F.o_xyz[(size_t) t * 3 + 1] = y;
        F.o_xyz[(size_t) t * 3 + 2] = z;
    }
    if (F.o_int)
        F.o_int[t] = inten;
    if (F.o_orig)
        F.o_orig[t] = src;
}

__global__ __launch_bounds__(256) void k_kitti_transpose_cells(const int* __restrict__ cell_src, int* __restrict__ out)
{
    // [2200][64] -> the reference's [64][2200] for cc_kitti_frame_result
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < ROWS * COLS)
        out[(t & (ROWS - 1)) * COLS + (t >> 6)] = cell_src[t];
}

} // namespace

struct cc_kitti
{
    int device = 0;
    int max_frames = 0;
    long long max_points = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // one allocation per kind, sliced per slot
    float4* d_in = nullptr;
    float4* d_pts = nullptr;
    unsigned char* d_laser = nullptr;
    unsigned short* d_col = nullptr;
    int* d_order = nullptr;
    unsigned short* d_ocol = nullptr;
    int* d_cell = nullptr;
    int* d_hist = nullptr;
    long long* d_info = nullptr;
    double* d_bins = nullptr;
    int* d_cell_t = nullptr; // transposed read-back scratch
    FrameSlot* d_slots = nullptr;
    std::vector<FrameSlot> h_slots;
    int last_frames = 0;
    static constexpr int MAX_BINS = 512;
};

extern "C" {

const char* cc_kitti_last_error(void)
{
    return g_kitti_error.c_str();
}

int cc_kitti_create(cc_kitti** out, int device, int max_frames, int64_t max_points, void* hip_stream)
{
    if (!out || max_frames <= 0 || max_points <= 0)
        return fail(CC_ERR_INVALID_ARGUMENT, "cc_kitti_create: bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return fail(CC_ERR_NO_DEVICE, "cc_kitti_create: no gfx950 device (there is no CPU variant of this path)");
    KITTI_HIP_CHECK(hipSetDevice(device));
    cc_kitti* k = new cc_kitti;
    k->device = device;
    k->max_frames = max_frames;
    k->max_points = (max_points + 1023) / 1024 * 1024;
    if (hip_stream)
        k->stream = (hipStream_t) hip_stream;
    else
    {
        KITTI_HIP_CHECK(hipStreamCreateWithFlags(&k->stream, hipStreamNonBlocking));
        k->own_stream = true;
    }
    const size_t F = (size_t) max_frames, N = (size_t) k->max_points;
    KITTI_HIP_CHECK(hipMalloc(&k->d_in, F * N * sizeof(float4)));
    KITTI_HIP_CHECK(hipMalloc(&k->d_pts, F * N * sizeof(float4)));
    KITTI_HIP_CHECK(hipMalloc(&k->d_laser, F * N));
    KITTI_HIP_CHECK(hipMalloc(&k->d_col, F * N * 2));
    KITTI_HIP_CHECK(hipMalloc(&k->d_order, F * N * 4));
    KITTI_HIP_CHECK(hipMalloc(&k->d_ocol, F * N * 2));
    KITTI_HIP_CHECK(hipMalloc(&k->d_cell, F * ROWS * COLS * 4));
    KITTI_HIP_CHECK(hipMalloc(&k->d_hist, F * ROWS * 4));
    KITTI_HIP_CHECK(hipMalloc(&k->d_info, F * 4 * sizeof(long long)));
    KITTI_HIP_CHECK(hipMalloc(&k->d_bins, F * cc_kitti::MAX_BINS * 12 * sizeof(double)));
    KITTI_HIP_CHECK(hipMalloc(&k->d_cell_t, (size_t) ROWS * COLS * 4));
    KITTI_HIP_CHECK(hipMalloc(&k->d_slots, F * sizeof(FrameSlot)));
    k->h_slots.resize(F);
    *out = k;
    return CC_OK;
}

void cc_kitti_destroy(cc_kitti* k)
{
    if (!k)
        return;
    (void) hipSetDevice(k->device);
    (void) hipStreamSynchronize(k->stream);
    for (void* p : {(void*) k->d_in, (void*) k->d_pts, (void*) k->d_laser, (void*) k->d_col, (void*) k->d_order, (void*) k->d_ocol,
                    (void*) k->d_cell, (void*) k->d_hist, (void*) k->d_info, (void*) k->d_bins, (void*) k->d_cell_t, (void*) k->d_slots})
        if (p)
            (void) hipFree(p);
    if (k->own_stream)
        (void) hipStreamDestroy(k->stream);
    delete k;
}

void* cc_kitti_hip_stream(cc_kitti* k)
{
    return k ? (void*) k->stream : nullptr;
}

int cc_kitti_sync(cc_kitti* k)
{
    if (!k)
        return fail(CC_ERR_INVALID_ARGUMENT, "null handle");
    KITTI_HIP_CHECK(hipSetDevice(k->device));
    KITTI_HIP_CHECK(hipStreamSynchronize(k->stream));
    return CC_OK;
}

int cc_kitti_convert_frames(cc_kitti* k, int n_frames, const cc_kitti_frame* frames)
{
    if (!k || n_frames < 0 || n_frames > k->max_frames || (n_frames > 0 && !frames))
        return fail(CC_ERR_INVALID_ARGUMENT, "cc_kitti_convert_frames: bad frame count");
    KITTI_HIP_CHECK(hipSetDevice(k->device));
    k->last_frames = n_frames;
    if (n_frames == 0)
        return CC_OK;
    const size_t N = (size_t) k->max_points;
    // the slot table of the previous call may still be read by kernels in flight
    KITTI_HIP_CHECK(hipStreamSynchronize(k->stream));
    for (int f = 0; f < n_frames; f++)
    {
        const cc_kitti_frame& fr = frames[f];
        if (fr.n_points < 0 || fr.n_points > k->max_points || (fr.n_points > 0 && !fr.points))
            return fail(CC_ERR_INVALID_ARGUMENT, "cc_kitti_convert_frames: frame " + std::to_string(f) + " has too many points or no buffer");
        if ((fr.stages & CC_KITTI_FIRINGS) && !(fr.stages & CC_KITTI_RANGE_IMAGE))
            return fail(CC_ERR_INVALID_ARGUMENT, "CC_KITTI_FIRINGS needs CC_KITTI_RANGE_IMAGE");
        if ((fr.stages & CC_KITTI_UNDO_EGO_MOTION) && (fr.num_bins <= 0 || fr.num_bins > cc_kitti::MAX_BINS || !fr.bin_transforms))
            return fail(CC_ERR_INVALID_ARGUMENT, "CC_KITTI_UNDO_EGO_MOTION needs 1.." + std::to_string(cc_kitti::MAX_BINS) + " bin transforms");
        if ((fr.stages & CC_KITTI_UNDO_EGO_MOTION) && fr.rotation_end_stamp < fr.rotation_start_stamp)
            return fail(CC_ERR_INVALID_ARGUMENT, "rotation_end_stamp < rotation_start_stamp");
        FrameSlot& S = k->h_slots[f];
        S.pts_in = k->d_in + f * N;
        S.pts = k->d_pts + f * N;
        S.laser = k->d_laser + f * N;
        S.col = k->d_col + f * N;
        S.order = k->d_order + f * N;
        S.ocol = k->d_ocol + f * N;
        S.cell_src = k->d_cell + (size_t) f * ROWS * COLS;
        S.hist = k->d_hist + f * ROWS;
        S.info = k->d_info + f * 4;
        S.bins = k->d_bins + (size_t) f * cc_kitti::MAX_BINS * 12;
        S.n = fr.n_points;
        S.duration = (double) (fr.rotation_end_stamp - fr.rotation_start_stamp); // static_cast<double>(duration), :199-203
        S.num_bins = (fr.stages & CC_KITTI_UNDO_EGO_MOTION) ? fr.num_bins : 0;
        S.stages = fr.stages;
        S.o_xyz = fr.d_xyz;
        S.o_int = fr.d_intensity;
        S.o_orig = fr.d_original_index;
        if (fr.n_points > 0)
            KITTI_HIP_CHECK(hipMemcpyAsync((void*) S.pts_in, fr.points, (size_t) fr.n_points * sizeof(float4), hipMemcpyHostToDevice, k->stream));
        if (!(fr.stages & CC_KITTI_RECOVER_ROWS))
        {
            if (fr.laser_index && fr.n_points > 0)
                KITTI_HIP_CHECK(hipMemcpyAsync(S.laser, fr.laser_index, (size_t) fr.n_points, hipMemcpyHostToDevice, k->stream));
            else if (fr.n_points > 0)
                KITTI_HIP_CHECK(hipMemsetAsync(S.laser, 0, (size_t) fr.n_points, k->stream));
        }
        if (S.num_bins > 0)
            KITTI_HIP_CHECK(hipMemcpyAsync((void*) S.bins, fr.bin_transforms, (size_t) fr.num_bins * 12 * sizeof(double), hipMemcpyHostToDevice, k->stream));
    }
    KITTI_HIP_CHECK(hipMemcpyAsync(k->d_slots, k->h_slots.data(), (size_t) n_frames * sizeof(FrameSlot), hipMemcpyHostToDevice, k->stream));
    KITTI_HIP_CHECK(hipMemsetAsync(k->d_cell, 0xFF, (size_t) n_frames * ROWS * COLS * 4, k->stream));
    KITTI_HIP_CHECK(hipMemsetAsync(k->d_info, 0, (size_t) n_frames * 4 * sizeof(long long), k->stream));
    hipLaunchKernelGGL(k_kitti_points, dim3(n_frames), dim3(1024), 0, k->stream, k->d_slots);
    hipLaunchKernelGGL(k_kitti_image, dim3(n_frames), dim3(64), 0, k->stream, k->d_slots);
    hipLaunchKernelGGL(k_kitti_firings, dim3((ROWS * COLS + 255) / 256, n_frames), dim3(256), 0, k->stream, k->d_slots);
    KITTI_HIP_CHECK(hipGetLastError());
    return CC_OK;
}

int cc_kitti_frame_result(cc_kitti* k, int slot, cc_kitti_frame_info* info, float* h_points, uint8_t* h_laser_index, int32_t* h_cell_source)
{
    if (!k || slot < 0 || slot >= k->last_frames)
        return fail(CC_ERR_INVALID_ARGUMENT, "cc_kitti_frame_result: no such slot in the last call");
    KITTI_HIP_CHECK(hipSetDevice(k->device));
    const FrameSlot& S = k->h_slots[slot];
    if (h_cell_source)
        hipLaunchKernelGGL(k_kitti_transpose_cells, dim3((ROWS * COLS + 255) / 256), dim3(256), 0, k->stream, S.cell_src, k->d_cell_t);
    KITTI_HIP_CHECK(hipStreamSynchronize(k->stream));
    if (info)
    {
        long long v[4];
        KITTI_HIP_CHECK(hipMemcpy(v, S.info, sizeof(v), hipMemcpyDeviceToHost));
        const long long jumps = v[0];
        info->rows_found = (int32_t) ((jumps < ROWS ? jumps : ROWS) + 1); // laser_index + 1 (:92), laser_index stops at 64
        info->break_index = v[1];
        info->max_columns = (int32_t) v[2];
        info->skipped = v[3];
    }
    if (h_points && S.n > 0)
        KITTI_HIP_CHECK(hipMemcpy(h_points, S.pts, (size_t) S.n * sizeof(float4), hipMemcpyDeviceToHost));
    if (h_laser_index && S.n > 0)
        KITTI_HIP_CHECK(hipMemcpy(h_laser_index, S.laser, (size_t) S.n, hipMemcpyDeviceToHost));
    if (h_cell_source)
        KITTI_HIP_CHECK(hipMemcpy(h_cell_source, k->d_cell_t, (size_t) ROWS * COLS * 4, hipMemcpyDeviceToHost));
    return CC_OK;
}

// ---- host pose arithmetic --------------------------------------------------------------------------------------------------
// 3x4 row-major [R|t]; products evaluated like Eigen's fixed-size coefficient products: sum over k in ascending order.

namespace
{
struct Iso
{
    double m[12];
    double r(int i, int j) const
    {
        return m[i * 4 + j];
    }
    double t(int i) const
    {
        return m[i * 4 + 3];
    }
};

Iso iso_mul(const Iso& a, const Iso& b) // Transform * Transform, Isometry mode (Eigen/src/Geometry/Transform.h)
{
    Iso c;
    for (int i = 0; i < 3; i++)
    {
        for (int j = 0; j < 3; j++)
            c.m[i * 4 + j] = (a.r(i, 0) * b.r(0, j) + a.r(i, 1) * b.r(1, j)) + a.r(i, 2) * b.r(2, j);
        c.m[i * 4 + 3] = ((a.r(i, 0) * b.t(0) + a.r(i, 1) * b.t(1)) + a.r(i, 2) * b.t(2)) + a.t(i);
    }
    return c;
}

Iso iso_inverse(const Iso& a) // Transform::inverse(Isometry): R^T, -R^T t
{
    Iso c;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            c.m[i * 4 + j] = a.r(j, i);
    for (int i = 0; i < 3; i++)
        c.m[i * 4 + 3] = ((-c.r(i, 0)) * a.t(0) + (-c.r(i, 1)) * a.t(1)) + (-c.r(i, 2)) * a.t(2);
    return c;
}

struct Quat
{
    double c[4]; // x, y, z, w (Eigen's coefficient order)
};

Quat quat_from_rotation(const Iso& a) // QuaternionBase::operator=(MatrixBase) for 3x3 (Eigen/src/Geometry/Quaternion.h)
{
    Quat q;
    double t = (a.r(0, 0) + a.r(1, 1)) + a.r(2, 2);
    if (t > 0.0)
    {
        t = std::sqrt(t + 1.0);
        q.c[3] = 0.5 * t;
        t = 0.5 / t;
        q.c[0] = (a.r(2, 1) - a.r(1, 2)) * t;
        q.c[1] = (a.r(0, 2) - a.r(2, 0)) * t;
        q.c[2] = (a.r(1, 0) - a.r(0, 1)) * t;
    }
    else
    {
        int i = 0;
        if (a.r(1, 1) > a.r(0, 0))
            i = 1;
        if (a.r(2, 2) > a.r(i, i))
            i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(a.r(i, i) - a.r(j, j) - a.r(k, k) + 1.0);
        q.c[i] = 0.5 * t;
        t = 0.5 / t;
        q.c[3] = (a.r(k, j) - a.r(j, k)) * t;
        q.c[j] = (a.r(j, i) + a.r(i, j)) * t;
        q.c[k] = (a.r(k, i) + a.r(i, k)) * t;
    }
    return q;
}

Quat quat_slerp(const Quat& a, double t, const Quat& b) // QuaternionBase::slerp
{
    const double one = 1.0 - 2.220446049250313e-16;
    // 4-coefficient dot product as two 2-wide packets: (x x' + z z') + (y y' + w w')
    const double d = (a.c[0] * b.c[0] + a.c[2] * b.c[2]) + (a.c[1] * b.c[1] + a.c[3] * b.c[3]);
    const double abs_d = std::fabs(d);
    double scale0, scale1;
    if (abs_d >= one)
    {
        scale0 = 1.0 - t;
        scale1 = t;
    }
    else
    {
        const double theta = std::acos(abs_d);
        const double sin_theta = std::sin(theta);
        scale0 = std::sin((1.0 - t) * theta) / sin_theta;
        scale1 = std::sin(t * theta) / sin_theta;
    }
    if (d < 0.0)
        scale1 = -scale1;
    Quat q;
    for (int i = 0; i < 4; i++)
        q.c[i] = scale0 * a.c[i] + scale1 * b.c[i];
    return q;
}

void quat_to_rotation(const Quat& q, Iso& out) // QuaternionBase::toRotationMatrix
{
    const double x = q.c[0], y = q.c[1], z = q.c[2], w = q.c[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    out.m[0] = 1.0 - (tyy + tzz);
    out.m[1] = txy - twz;
    out.m[2] = txz + twy;
    out.m[4] = txy + twz;
    out.m[5] = 1.0 - (txx + tzz);
    out.m[6] = tyz - twx;
    out.m[8] = txz - twy;
    out.m[9] = tyz + twx;
    out.m[10] = 1.0 - (txx + tyy);
}

Iso interpolate(int64_t n, const uint64_t* stamps, const double* poses, uint64_t stamp) // kitti_loader.cpp:297-328
{
    const uint64_t* after = std::lower_bound(stamps, stamps + n, stamp);
    const int64_t ia = after - stamps;
    Iso out;
    if (ia == n)
    {
        std::memcpy(out.m, poses + (n - 1) * 12, sizeof(out.m));
        return out;
    }
    if (ia == 0)
    {
        std::memcpy(out.m, poses, sizeof(out.m));
        return out;
    }
    Iso pb, pa;
    std::memcpy(pb.m, poses + (ia - 1) * 12, sizeof(pb.m));
    std::memcpy(pa.m, poses + ia * 12, sizeof(pa.m));
    const double f = static_cast<double>(stamp - stamps[ia - 1]) / static_cast<double>(stamps[ia] - stamps[ia - 1]);
    const Quat q = quat_slerp(quat_from_rotation(pb), f, quat_from_rotation(pa));
    quat_to_rotation(q, out);
    for (int i = 0; i < 3; i++)
        out.m[i * 4 + 3] = (1 - f) * pb.t(i) + f * pa.t(i);
    return out;
}
} // namespace

int cc_kitti_pose_interpolate(int64_t n_poses, const uint64_t* stamps, const double* poses, uint64_t stamp, double out[12])
{
    if (n_poses <= 0 || !stamps || !poses || !out)
        return fail(CC_ERR_INVALID_ARGUMENT, "cc_kitti_pose_interpolate: bad argument");
    const Iso r = interpolate(n_poses, stamps, poses, stamp);
    std::memcpy(out, r.m, sizeof(r.m));
    return CC_OK;
}

int cc_kitti_bin_transforms(int64_t n_poses, const uint64_t* stamps, const double* poses, uint64_t rotation_start_stamp,
                            uint64_t rotation_end_stamp, const double mid_pose[12], double* out, int32_t capacity, int32_t* num_bins)
{
    if (n_poses <= 0 || !stamps || !poses || !mid_pose || !out || !num_bins || rotation_end_stamp < rotation_start_stamp)
        return fail(CC_ERR_INVALID_ARGUMENT, "cc_kitti_bin_transforms: bad argument");
    const uint64_t bin_resolution = 1000000; // 1 ms (:184)
    const uint64_t duration = rotation_end_stamp - rotation_start_stamp;
    const int nb = (int) std::ceil(static_cast<double>(duration) / static_cast<double>(bin_resolution));
    *num_bins = nb;
    if (nb > capacity)
        return fail(CC_ERR_CAPACITY, "cc_kitti_bin_transforms: " + std::to_string(nb) + " bins do not fit");
    Iso mid;
    std::memcpy(mid.m, mid_pose, sizeof(mid.m));
    for (int b = 0; b < nb; b++)
    {
        const uint64_t stamp_at_bin = rotation_start_stamp + (uint64_t) b * bin_resolution + (bin_resolution / 2);
        const Iso r = iso_mul(iso_inverse(interpolate(n_poses, stamps, poses, stamp_at_bin)), mid);
        std::memcpy(out + (size_t) b * 12, r.m, sizeof(r.m));
    }
    return CC_OK;
}

int cc_kitti_firing_stamps_and_poses(int64_t n_poses, const uint64_t* stamps, const double* poses, uint64_t start_stamp,
                                     uint64_t end_stamp, uint64_t* out_stamps, double* out_poses)
{
    if (n_poses <= 0 || !stamps || !poses || end_stamp < start_stamp)
        return fail(CC_ERR_INVALID_ARGUMENT, "cc_kitti_firing_stamps_and_poses: bad argument");
    for (int c = 0; c < COLS; c++)
    {
        const double elapsed_ratio = static_cast<double>(c) / (COLS - 1);
        const double elapsed_time = static_cast<double>(end_stamp - start_stamp) * elapsed_ratio;
        const uint64_t stamp = start_stamp + static_cast<uint64_t>(elapsed_time);
        if (out_stamps)
            out_stamps[c] = stamp;
        if (out_poses)
        {
            const Iso r = interpolate(n_poses, stamps, poses, stamp);
            std::memcpy(out_poses + (size_t) c * 12, r.m, sizeof(r.m));
        }
    }
    return CC_OK;
}

int cc_kitti_start_end_stamps(int64_t n, const uint64_t* middle, uint64_t* start, uint64_t* end)
{
    if (n <= 0 || !middle || !start || !end)
        return fail(CC_ERR_INVALID_ARGUMENT, "cc_kitti_start_end_stamps: bad argument");
    for (int64_t i = 0; i + 1 < n; i++)
    {
        end[i] = (middle[i] + middle[i + 1]) / 2;
        start[i + 1] = end[i];
    }
    start[0] = middle[0] - 50000000UL;
    end[n - 1] = middle[n - 1] + 50000000UL;
    return CC_OK;
}

int cc_kitti_pose_from_line(const double row12[12], const double cam0_from_x[12], double out[12])
{
    if (!row12 || !cam0_from_x || !out)
        return fail(CC_ERR_INVALID_ARGUMENT, "cc_kitti_pose_from_line: bad argument");
    Iso odom_from_first_cam0 = {{0, 0, 1, 0, -1, 0, 0, 0, 0, -1, 0, 0}}; // kitti_loader.cpp:339-340
    Iso a, b;
    std::memcpy(a.m, row12, sizeof(a.m));
    std::memcpy(b.m, cam0_from_x, sizeof(b.m));
    const Iso r = iso_mul(iso_mul(odom_from_first_cam0, a), b);
    std::memcpy(out, r.m, sizeof(r.m));
    return CC_OK;
}

} // extern "C"
