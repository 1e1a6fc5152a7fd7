// cc_gt_labels.hip — ground-truth euclidean-clustering labels on gfx950.
//
// KittiEvaluation::generateEuclideanClusteringLabels (src/evaluation/kitti_evaluation.cpp:224-275) runs PCL's
// ConditionalEuclideanClustering over one KITTI frame: region growing from every not-yet-processed point in index order, a
// neighbour joins when its squared distance is < 1 m^2 and it carries the same semantic and instance label (isSameCluster,
// :270-275); clusters with 10..300000 points are kept and numbered 1, 2, ... in the order their first point appears; points of the
// ground / unlabeled classes get 0 but their clusters still consume a number (:253-262).
//
// Region growing that only marks points it accepts explores whole connected components, so the clusters are the connected
// components of the graph {d^2 < 1, same labels} — an order-free definition the GPU can build in parallel:
//
//   k_gt_cells    hash every point's 1-m grid cell into an open-addressing table (atomicCAS on the packed cell key) and push the
//                 point on the cell's list (atomicExch on the head)
//   k_gt_link     per point: walk the 27 neighbouring cells' lists, and for every earlier point within range with equal labels
//                 union the two in a lock-free union-find (atomicCAS on parent, smaller index wins => the root of a component is its
//                 first point). The distance is FLANN's L2_Simple<float>: ((dx*dx) + dy*dy) + dz*dz in float, no FMA.
//   k_gt_roots    flatten, count component sizes
//   k_gt_rank     one block: prefix sum over the points of "is the root of a kept component" => cluster number in first-point order
//   k_gt_labels   label = ground/unlabeled class ? 0 : number of the point's component (0 for dropped components)
//
// A pair closer than 1 m differs by less than 1 m per axis, hence lies in adjacent floor() cells, also after float rounding (if the
// exact |dx| >= 1 then fl(dx)^2 >= 1 and the float sum cannot be < 1).
// PARITY UNPINNED with respect to PCL itself (not in this image); see oracle/gt_oracle.cpp for the sequential restatement.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/cc_hip.h"

namespace
{

constexpr unsigned long long EMPTY_KEY = ~0ull;
constexpr int CELL_BIAS = 1 << 20;
constexpr int MIN_CLUSTER_SIZE = 10, MAX_CLUSTER_SIZE = 300000; // kitti_evaluation.hpp:53-57
constexpr uint16_t L_UNLABELED = 0, L_ROAD = 40, L_PARKING = 44, L_SIDEWALK = 48, L_OTHER_GROUND = 49, L_LANE_MARKING = 60, L_TERRAIN = 72;

struct Gt
{
    const float4* pts;
    const uint32_t* lab; // semantic | instance << 16
    long long n;
    unsigned mask;
    unsigned long long* keys;
    int* head;   // per table slot: last point pushed, -1 = none
    int* next;   // per point
    int* parent; // union-find
    int* size;   // per root
    int* rank;   // per root: cluster number (0 = dropped)
    uint16_t* out;
};

__device__ __forceinline__ bool cell_of(const float4& p, int& cx, int& cy, int& cz)
{
    if (!(fabsf(p.x) < 1.0e6f && fabsf(p.y) < 1.0e6f && fabsf(p.z) < 1.0e6f)) // NaN / inf / absurd: the point has no neighbours
        return false;
    cx = (int) floorf(p.x);
    cy = (int) floorf(p.y);
    cz = (int) floorf(p.z);
    return true;
}

__device__ __forceinline__ unsigned long long pack_cell(int cx, int cy, int cz)
{
    return ((unsigned long long) (cx + CELL_BIAS) << 42) | ((unsigned long long) (cy + CELL_BIAS) << 21) | (unsigned long long) (cz + CELL_BIAS);
}

__device__ __forceinline__ unsigned slot_of(unsigned long long key, unsigned mask)
{
    return (unsigned) ((key * 0x9E3779B97F4A7C15ull) >> 38) & mask;
}

__global__ __launch_bounds__(256) void k_gt_cells(Gt g)
{
    const long long i = (long long) blockIdx.x * 256 + threadIdx.x;
    if (i >= g.n)
        return;
    g.parent[i] = (int) i;
    g.size[i] = 0;
    g.rank[i] = 0;
    g.next[i] = -1;
    int cx, cy, cz;
    if (!cell_of(g.pts[i], cx, cy, cz))
        return;
    const unsigned long long key = pack_cell(cx, cy, cz);
    unsigned s = slot_of(key, g.mask);
    while (true)
    {
        const unsigned long long old = atomicCAS(&g.keys[s], EMPTY_KEY, key);
        if (old == EMPTY_KEY || old == key)
            break;
        s = (s + 1) & g.mask;
    }
    g.next[i] = atomicExch(&g.head[s], (int) i);
}

__device__ __forceinline__ int uf_find(int* parent, int x)
{
    while (true)
    {
        const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x)
            return x;
        x = p;
    }
}

__device__ __forceinline__ void uf_union(int* parent, int a, int b)
{
    while (true)
    {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b)
            return;
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        if (atomicCAS(&parent[hi], hi, lo) == hi) // the smaller index becomes the root
            return;
    }
}

__global__ __launch_bounds__(256) void k_gt_link(Gt g)
{
    const long long i = (long long) blockIdx.x * 256 + threadIdx.x;
    if (i >= g.n)
        return;
    const float4 p = g.pts[i];
    int cx, cy, cz;
    if (!cell_of(p, cx, cy, cz))
        return;
    const uint32_t lab = g.lab[i];
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++)
            {
                const unsigned long long key = pack_cell(cx + dx, cy + dy, cz + dz);
                unsigned s = slot_of(key, g.mask);
                int q = -1;
                while (true)
                {
                    const unsigned long long k = g.keys[s];
                    if (k == key)
                    {
                        q = g.head[s];
                        break;
                    }
                    if (k == EMPTY_KEY)
                        break;
                    s = (s + 1) & g.mask;
                }
                for (; q >= 0; q = g.next[q])
                {
                    if (q >= i || g.lab[q] != lab) // every pair once; isSameCluster: equal semantic and instance label
                        continue;
                    const float4 o = g.pts[q];
                    const float ex = p.x - o.x, ey = p.y - o.y, ez = p.z - o.z;
                    const float d2 = (ex * ex + ey * ey) + ez * ez; // L2_Simple<float>; -ffp-contract=off
                    if (d2 < 1.0f * 1.0f)
                        uf_union(g.parent, (int) i, q);
                }
            }
}

__global__ __launch_bounds__(256) void k_gt_roots(Gt g)
{
    const long long i = (long long) blockIdx.x * 256 + threadIdx.x;
    if (i >= g.n)
        return;
    const int r = uf_find(g.parent, (int) i);
    g.next[i] = r; // the cell lists are no longer needed: reuse as "root of point"
    atomicAdd(&g.size[r], 1);
}

__global__ __launch_bounds__(1024) void k_gt_rank(Gt g)
{
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0)
        s_carry = 0;
    __syncthreads();
    for (long long base = 0; base < g.n; base += 1024)
    {
        const long long i = base + tid;
        int keep = 0;
        if (i < g.n && g.next[i] == (int) i)
        {
            const int sz = g.size[i];
            keep = sz >= MIN_CLUSTER_SIZE && sz <= MAX_CLUSTER_SIZE;
        }
        int v = keep;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1)
        {
            const int o = __shfl_up(v, d, 64);
            if (lane >= d)
                v += o;
        }
        if (lane == 63)
            s_w[wid] = v;
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wid; w++)
            before += s_w[w];
        if (keep)
            g.rank[i] = before + v; // 1-based cluster_index (:251)
        __syncthreads();
        if (tid == 1023)
            s_carry = before + v;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_gt_labels(Gt g)
{
    const long long i = (long long) blockIdx.x * 256 + threadIdx.x;
    if (i >= g.n)
        return;
    const uint16_t s = (uint16_t) (g.lab[i] & 0xffff);
    const bool zero = s == L_LANE_MARKING || s == L_ROAD || s == L_PARKING || s == L_SIDEWALK || s == L_OTHER_GROUND || s == L_TERRAIN ||
                      s == L_UNLABELED; // :256-259
    g.out[i] = zero ? (uint16_t) 0 : (uint16_t) g.rank[g.next[i]];
}

} // namespace

extern "C" int cc_eval_generate_euclidean_labels(int device, int64_t n, const float* points, const uint16_t* semantic, const uint16_t* instance,
                                                 uint16_t* out_labels)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return CC_ERR_NO_DEVICE;
    if (n < 0 || n > (1ll << 30) || (n > 0 && (!points || !semantic || !instance || !out_labels)))
        return CC_ERR_INVALID_ARGUMENT;
    if (n == 0)
        return CC_OK;
    if (hipSetDevice(device) != hipSuccess)
        return CC_ERR_HIP;
    unsigned table = 1024;
    while ((int64_t) table < 2 * n)
        table <<= 1;
    // one allocation: points | labels | keys | head | next | parent | size | rank | out
    const size_t N = (size_t) n;
    size_t off = 0;
    auto take = [&](size_t bytes)
    {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t) 255;
        return o;
    };
    const size_t o_pts = take(N * 16), o_lab = take(N * 4), o_keys = take((size_t) table * 8), o_head = take((size_t) table * 4),
                 o_next = take(N * 4), o_parent = take(N * 4), o_size = take(N * 4), o_rank = take(N * 4), o_out = take(N * 2);
    char* d = nullptr;
    if (hipMalloc(&d, off) != hipSuccess)
        return CC_ERR_HIP;
    std::vector<uint32_t> lab(N);
    for (size_t i = 0; i < N; i++)
        lab[i] = (uint32_t) semantic[i] | ((uint32_t) instance[i] << 16);
    int rc = CC_ERR_HIP;
    if (!hipMemcpy(d + o_pts, points, N * 16, hipMemcpyHostToDevice) && !hipMemcpy(d + o_lab, lab.data(), N * 4, hipMemcpyHostToDevice) &&
        !hipMemset(d + o_keys, 0xFF, (size_t) table * 8) && !hipMemset(d + o_head, 0xFF, (size_t) table * 4))
    {
        Gt g;
        g.pts = (const float4*) (d + o_pts);
        g.lab = (const uint32_t*) (d + o_lab);
        g.n = n;
        g.mask = table - 1;
        g.keys = (unsigned long long*) (d + o_keys);
        g.head = (int*) (d + o_head);
        g.next = (int*) (d + o_next);
        g.parent = (int*) (d + o_parent);
        g.size = (int*) (d + o_size);
        g.rank = (int*) (d + o_rank);
        g.out = (uint16_t*) (d + o_out);
        const unsigned blocks = (unsigned) ((n + 255) / 256);
        hipLaunchKernelGGL(k_gt_cells, dim3(blocks), dim3(256), 0, 0, g);
        hipLaunchKernelGGL(k_gt_link, dim3(blocks), dim3(256), 0, 0, g);
        hipLaunchKernelGGL(k_gt_roots, dim3(blocks), dim3(256), 0, 0, g);
        hipLaunchKernelGGL(k_gt_rank, dim3(1), dim3(1024), 0, 0, g);
        hipLaunchKernelGGL(k_gt_labels, dim3(blocks), dim3(256), 0, 0, g);
        if (hipGetLastError() == hipSuccess && hipMemcpy(out_labels, d + o_out, N * 2, hipMemcpyDeviceToHost) == hipSuccess)
            rc = CC_OK;
    }
    (void) hipFree(d);
    return rc;
}
