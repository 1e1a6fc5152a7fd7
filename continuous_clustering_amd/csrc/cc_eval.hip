// cc_eval.hip — label compare of the evaluation path (src/evaluation/kitti_evaluation.cpp:29-146, 277-293) on gfx950.
//
// GPU part (exact integer work, order independent): ground confusion counts (evaluateGroundPoints, :44-84) and the
// contingency table between ground-truth euclidean-clustering labels and detection ids (what evaluateClusters builds with nested
// std::maps, :86-146) through a lock-free atomicCAS hash table over the 64-bit key (gt << 32 | detection).
// Host part: the entropies are summed from the exact integer table in the reference's order (ascending std::map keys, outer
// then inner) with the host's std::log, so the result is bit-identical to the reference's on the same host — no floating-point
// reduction happens on the device.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
// RCCL is not a build-time dependency: the three types cc_eval_gather_records names are declared here as rccl.h declares them (ncclComm_t is an
// opaque pointer, ncclDouble = 8 and ncclSuccess = 0 in every NCCL / RCCL release), the library itself is resolved in the process at run time
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
#include <mutex>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/cc_hip.h"

namespace
{

// semantic label ids of KittiLoader::getSemanticKittiLabelNameToLabelNumericMapping (kitti_loader.cpp:566-604) used by
// KittiEvaluation (kitti_evaluation.cpp:12-27)
constexpr uint16_t LABEL_UNLABELED = 0, LABEL_ROAD = 40, LABEL_PARKING = 44, LABEL_SIDEWALK = 48, LABEL_OTHER_GROUND = 49,
                   LABEL_LANE_MARKING = 60, LABEL_TERRAIN = 72;
constexpr unsigned long long EMPTY = ~0ull;

__global__ __launch_bounds__(256) void k_eval(long long n, const uint16_t* __restrict__ semantic, const uint32_t* __restrict__ euclid,
                                              const uint8_t* __restrict__ is_ground, const uint32_t* __restrict__ detection,
                                              unsigned long long* counts4, unsigned long long* keys, unsigned* vals, unsigned mask)
{
    const long long i = (long long) blockIdx.x * 256 + threadIdx.x;
    unsigned tp = 0, fn = 0, fp = 0, tn = 0;
    if (i < n)
    {
        const uint16_t sl = semantic[i];
        if (sl != LABEL_UNLABELED) // kitti_evaluation.cpp:49-50
        {
            const bool gt_ground = sl == LABEL_LANE_MARKING || sl == LABEL_ROAD || sl == LABEL_PARKING || sl == LABEL_SIDEWALK ||
                                   sl == LABEL_OTHER_GROUND || sl == LABEL_TERRAIN;
            const bool seg_ground = is_ground[i] != 0;
            tp = gt_ground && seg_ground;
            fn = gt_ground && !seg_ground;
            fp = !gt_ground && seg_ground;
            tn = !gt_ground && !seg_ground;
        }
        const uint32_t gt = euclid[i], det = detection[i];
        if (gt != 0 || det != 0) // points with neither label take part in no entropy term (kitti_evaluation.cpp:93-99)
        {
            const unsigned long long key = ((unsigned long long) gt << 32) | det;
            unsigned long long h = key * 0x9E3779B97F4A7C15ull;
            unsigned slot = (unsigned) (h >> 40) & mask;
            if (key == EMPTY) // (the one pair that looks like a free slot is counted beside the table)
                atomicAdd(&counts4[4], 1ull);
            else
            while (true)
            {
                const unsigned long long old = atomicCAS(&keys[slot], EMPTY, key);
                if (old == EMPTY || old == key)
                {
                    atomicAdd(&vals[slot], 1u);
                    break;
                }
                slot = (slot + 1) & mask;
            }
        }
    }
    // wave-level reduction of the four counters, one atomic per wave
    const unsigned long long m_tp = __ballot(tp), m_fn = __ballot(fn), m_fp = __ballot(fp), m_tn = __ballot(tn);
    if ((threadIdx.x & 63) == 0)
    {
        if (m_tp) atomicAdd(&counts4[0], (unsigned long long) __popcll(m_tp));
        if (m_fn) atomicAdd(&counts4[1], (unsigned long long) __popcll(m_fn));
        if (m_fp) atomicAdd(&counts4[2], (unsigned long long) __popcll(m_fp));
        if (m_tn) atomicAdd(&counts4[3], (unsigned long long) __popcll(m_tn));
    }
}

__global__ __launch_bounds__(256) void k_eval_compact(unsigned table, const unsigned long long* keys, const unsigned* vals,
                                                      unsigned long long* out_keys, unsigned* out_vals, unsigned* out_n)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < table && keys[i] != EMPTY)
    {
        const unsigned o = atomicAdd(out_n, 1u);
        out_keys[o] = keys[i];
        out_vals[o] = vals[i];
    }
}

struct Pair
{
    uint32_t gt, det, n;
};

} // namespace

extern "C" {

int cc_eval_frame_device(int64_t n, const uint16_t* d_semantic, const uint32_t* d_euclid, const uint8_t* d_is_ground,
                         const uint32_t* d_detection, cc_eval_frame_result* out)
{
    if (n < 0 || !out || (n > 0 && (!d_semantic || !d_euclid || !d_is_ground || !d_detection)))
        return CC_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    if (n == 0)
        return CC_OK;
    if (n > (1ll << 30)) // (the table holds at least two slots per point: open addressing never meets a full table)
        return CC_ERR_INVALID_ARGUMENT;
    unsigned table = 1024;
    while ((int64_t) table < 2 * n)
        table <<= 1;
    unsigned long long *d_keys = nullptr, *d_counts = nullptr, *d_okeys = nullptr;
    unsigned *d_vals = nullptr, *d_ovals = nullptr, *d_on = nullptr;
    auto fail = [&]()
    {
        (void) hipFree(d_keys), (void) hipFree(d_counts), (void) hipFree(d_okeys), (void) hipFree(d_vals), (void) hipFree(d_ovals),
            (void) hipFree(d_on);
        return CC_ERR_HIP;
    };
    if (hipMalloc(&d_keys, (size_t) table * 8) || hipMalloc(&d_vals, (size_t) table * 4) || hipMalloc(&d_counts, 40) ||
        hipMalloc(&d_okeys, (size_t) table * 8) || hipMalloc(&d_ovals, (size_t) table * 4) || hipMalloc(&d_on, 4))
        return fail();
    if (hipMemset(d_keys, 0xFF, (size_t) table * 8) || hipMemset(d_vals, 0, (size_t) table * 4) || hipMemset(d_counts, 0, 40) ||
        hipMemset(d_on, 0, 4))
        return fail();
    hipLaunchKernelGGL(k_eval, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, (long long) n, d_semantic, d_euclid, d_is_ground,
                       d_detection, d_counts, d_keys, d_vals, table - 1);
    if (hipGetLastError() != hipSuccess)
        return fail();
    hipLaunchKernelGGL(k_eval_compact, dim3((table + 255) / 256), dim3(256), 0, 0, table, d_keys, d_vals, d_okeys, d_ovals, d_on);
    if (hipGetLastError() != hipSuccess)
        return fail();
    unsigned long long counts[5];
    unsigned npairs = 0;
    if (hipMemcpy(counts, d_counts, 40, hipMemcpyDeviceToHost) || hipMemcpy(&npairs, d_on, 4, hipMemcpyDeviceToHost))
        return fail();
    std::vector<unsigned long long> hk(npairs);
    std::vector<unsigned> hv(npairs);
    if (npairs && (hipMemcpy(hk.data(), d_okeys, (size_t) npairs * 8, hipMemcpyDeviceToHost) ||
                   hipMemcpy(hv.data(), d_ovals, (size_t) npairs * 4, hipMemcpyDeviceToHost)))
        return fail();
    (void) hipFree(d_keys), (void) hipFree(d_counts), (void) hipFree(d_okeys), (void) hipFree(d_vals), (void) hipFree(d_ovals),
        (void) hipFree(d_on);

    out->tp = (double) counts[0];
    out->fn = (double) counts[1];
    out->fp = (double) counts[2];
    out->tn = (double) counts[3];
    std::vector<Pair> pairs(npairs);
    for (unsigned i = 0; i < npairs; i++)
        pairs[i] = Pair{(uint32_t) (hk[i] >> 32), (uint32_t) hk[i], hv[i]};
    if (counts[4])
        pairs.push_back(Pair{0xffffffffu, 0xffffffffu, (unsigned) counts[4]});
    // over-segmentation entropy (kitti_evaluation.cpp:102-116): ground-truth clusters != 0 in ascending order, inside each the
    // detections (0 included) in ascending order
    std::sort(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) { return a.gt != b.gt ? a.gt < b.gt : a.det < b.det; });
    for (size_t i = 0; i < pairs.size();)
    {
        size_t j = i;
        unsigned long long total = 0;
        while (j < pairs.size() && pairs[j].gt == pairs[i].gt)
            total += pairs[j++].n;
        if (pairs[i].gt != 0)
            for (size_t k = i; k < j; k++)
            {
                const double frac = static_cast<double>(pairs[k].n) / static_cast<double>(total);
                out->over_segmentation_entropy -= frac * std::log(frac);
            }
        i = j;
    }
    // under-segmentation entropy (kitti_evaluation.cpp:125-145): detections != 0 ascending, inside each the ground-truth labels
    // (0 included) ascending; a detection that only contains ground truth 0 is skipped
    std::sort(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) { return a.det != b.det ? a.det < b.det : a.gt < b.gt; });
    for (size_t i = 0; i < pairs.size();)
    {
        size_t j = i;
        unsigned long long total = 0;
        while (j < pairs.size() && pairs[j].det == pairs[i].det)
            total += pairs[j++].n;
        const bool only_unlabeled = (j - i == 1) && pairs[i].gt == 0;
        if (pairs[i].det != 0 && !only_unlabeled)
            for (size_t k = i; k < j; k++)
            {
                const double frac = static_cast<double>(pairs[k].n) / static_cast<double>(total);
                out->under_segmentation_entropy -= frac * std::log(frac);
            }
        i = j;
    }
    return CC_OK;
}

int cc_eval_frame(int device, int64_t n, const uint16_t* semantic, const uint32_t* euclid, const uint8_t* is_ground,
                  const uint32_t* detection, cc_eval_frame_result* out)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return CC_ERR_NO_DEVICE;
    if (n < 0 || !out || (n > 0 && (!semantic || !euclid || !is_ground || !detection)))
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(device);
    if (n == 0)
    {
        memset(out, 0, sizeof(*out));
        return CC_OK;
    }
    uint16_t* d_s = nullptr;
    uint32_t *d_e = nullptr, *d_d = nullptr;
    uint8_t* d_g = nullptr;
    int rc = CC_ERR_HIP;
    if (!hipMalloc(&d_s, n * 2) && !hipMalloc(&d_e, n * 4) && !hipMalloc(&d_g, n) && !hipMalloc(&d_d, n * 4) &&
        !hipMemcpy(d_s, semantic, n * 2, hipMemcpyHostToDevice) && !hipMemcpy(d_e, euclid, n * 4, hipMemcpyHostToDevice) &&
        !hipMemcpy(d_g, is_ground, n, hipMemcpyHostToDevice) && !hipMemcpy(d_d, detection, n * 4, hipMemcpyHostToDevice))
        rc = cc_eval_frame_device(n, d_s, d_e, d_g, d_d, out);
    (void) hipFree(d_s), (void) hipFree(d_e), (void) hipFree(d_g), (void) hipFree(d_d);
    return rc;
}

// calculateMeanAndStdDev (kitti_evaluation.cpp:277-293): two-pass mean and population standard deviation, in input order.
void cc_eval_mean_std(const double* data, int64_t n, double* mean, double* std_dev)
{
    double m = 0;
    for (int64_t i = 0; i < n; i++)
        m += data[i];
    m /= static_cast<double>(n);
    double s = 0;
    for (int64_t i = 0; i < n; i++)
    {
        const double diff = data[i] - m;
        s += diff * diff;
    }
    *mean = m;
    *std_dev = std::sqrt(s / static_cast<double>(n));
}

// The six per-sequence metrics of generateEvaluationResults (kitti_evaluation.cpp:187-208): recall, precision, F1, accuracy
// (fractions, the reference prints them x100), USE, OSE; out = 6 x {mean, sigma}.
void cc_eval_summarize(const cc_eval_frame_result* frames, int64_t n, double out[12])
{
    std::vector<double> data((size_t) n);
    for (int metric = 0; metric < 6; metric++)
    {
        for (int64_t i = 0; i < n; i++)
        {
            const cc_eval_frame_result& r = frames[i];
            double v = 0;
            switch (metric)
            {
                case 0: v = r.tp / (r.tp + r.fn); break;
                case 1: v = r.tp / (r.tp + r.fp); break;
                case 2: v = (r.tp + r.tp) / (r.tp + r.tp + r.fp + r.fn); break;
                case 3: v = (r.tp + r.tn) / (r.tp + r.tn + r.fp + r.fn); break;
                case 4: v = r.under_segmentation_entropy; break;
                default: v = r.over_segmentation_entropy; break;
            }
            data[(size_t) i] = v;
        }
        cc_eval_mean_std(data.data(), n, &out[metric * 2], &out[metric * 2 + 1]);
    }
}


// ---- the one exchange step of the whole system: per-frame evaluation records of every rank to every rank -----------------
// RCCL is not a link-time dependency of this library: the collective is looked up in the librccl the process already has (the harness that
// made the communicator linked it), so nothing changes for callers that never gather.
static void* rccl_symbol(const char* name)
{
    static void* lib = nullptr;
    static std::once_flag once; // (callers may gather from several threads)
    std::call_once(once,
                   []()
                   {
                       for (const char* n : {"librccl.so.1", "librccl.so"})
                       {
                           lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); // the instance the communicator came from
                           if (lib)
                               return;
                       }
                       for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                       {
                           lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                           if (lib)
                               return;
                       }
                   });
    return lib ? dlsym(lib, name) : nullptr;
}

int cc_eval_gather_records(void* nccl_comm, int world, int device, const double* records, int64_t n, int64_t capacity, double* out,
                           int64_t* counts)
{
    if (!nccl_comm || world < 1 || n < 0 || capacity < 1 || n > capacity || (n > 0 && !records) || !out || !counts)
        return CC_ERR_INVALID_ARGUMENT;
    typedef ncclResult_t (*allgather_fn)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    static allgather_fn all_gather = nullptr;
    static std::once_flag once;
    std::call_once(once, []() { all_gather = (allgather_fn) rccl_symbol("ncclAllGather"); });
    if (!all_gather)
        return CC_ERR_HIP; // (no RCCL in this process)
    if (hipSetDevice(device) != hipSuccess)
        return CC_ERR_NO_DEVICE;
    // fixed-size padded block per rank: row 0 = {number of records}, rows 1 .. capacity = records of 8 doubles (sequence, frame, tp, fn, fp,
    // tn, OSE, USE: EvaluationResultForFrame, kitti_evaluation.hpp:38-49, tagged) — ONE ncclAllGather
    const size_t row = 8, block = (size_t) (capacity + 1) * row;
    std::vector<double> h_send(block, 0.0), h_recv(block * (size_t) world);
    h_send[0] = (double) n;
    if (n > 0)
        std::memcpy(h_send.data() + row, records, (size_t) n * row * sizeof(double));
    double *d_send = nullptr, *d_recv = nullptr;
    hipStream_t st = nullptr;
    int rc = CC_OK;
    if (hipMalloc(&d_send, block * sizeof(double)) != hipSuccess || hipMalloc(&d_recv, block * world * sizeof(double)) != hipSuccess ||
        hipStreamCreate(&st) != hipSuccess)
        rc = CC_ERR_HIP;
    if (rc == CC_OK && hipMemcpyAsync(d_send, h_send.data(), block * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess)
        rc = CC_ERR_HIP;
    if (rc == CC_OK && all_gather(d_send, d_recv, block, ncclDouble, (ncclComm_t) nccl_comm, st) != ncclSuccess)
        rc = CC_ERR_HIP;
    if (rc == CC_OK && (hipMemcpyAsync(h_recv.data(), d_recv, block * world * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess))
        rc = CC_ERR_HIP;
    if (rc == CC_OK)
        for (int r = 0; r < world; r++)
        {
            const double* b = h_recv.data() + (size_t) r * block;
            counts[r] = (int64_t) b[0];
            std::memcpy(out + (size_t) r * (size_t) capacity * row, b + row, (size_t) capacity * row * sizeof(double));
        }
    if (st)
        (void) hipStreamDestroy(st);
    if (d_send)
        (void) hipFree(d_send);
    if (d_recv)
        (void) hipFree(d_recv);
    return rc;
}

} // extern "C"
