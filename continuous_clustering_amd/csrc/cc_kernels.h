// cc_kernels.h — hand-written gfx950 kernels of the continuous-clustering hot path.
//
//   k_insert     one wavefront per sensor stream, lanes = rows of a firing (continuous range-image insertion,
//                continuous_clustering.cpp:105-292) — serial over the firings of a batch, parallel over lasers.
//   k_segment    one lane per finished column (ground-point segmentation + ignore flags, cc.cpp:294-624) —
//                parallel over all (stream, column) pairs of the batch.
//   k_associate  one wavefront per sensor stream, lanes = rows of a column: neighbour window scan + distance test
//                (cc.cpp:638-835), lock-free atomicCAS union-find over point trees, finished-cluster check and
//                column publishing / clearing (cc.cpp:837-1145) — serial over the columns of a batch.
//   k_view       gathers columns of one stream into the host-view layout (cc_engine_read_columns).
//
// All floating-point expressions keep the reference's operation order; the translation unit is compiled with
// -ffp-contract=off (and the pragma below) so that nothing is fused into FMAs the x86 reference does not have.
#pragma once
#include <hip/hip_runtime.h>

#include "cc_device.h"
#include "cc_math.h"

#pragma clang fp contract(off)

namespace cck
{
using namespace ccd;

#define CC_PI_F 3.14159274101257324219f  /* static_cast<float>(M_PI) */
#define CC_2PI_D 6.283185307179586       /* 2 * M_PI */

__device__ __forceinline__ int lane_id()
{
    return threadIdx.x & 63;
}

// static_cast<int>(float) as x86 cvttss2si does it: out of range / NaN -> INT_MIN
__device__ __forceinline__ int f2i_x86(float v)
{
    if (!(v > -2147483904.0f && v < 2147483648.0f))
        return (int) 0x80000000;
    return (int) v;
}

// Wave-wide reductions over all 64 lanes by DPP (row_shr 1/2/4/8 inside the rows of 16, then row_bcast 15 and 31): ~30 VALU
// instructions and no LDS traffic, where the ds_bpermute butterfly costs a lone wave twelve LDS round trips (~700 cycles).
// All lanes must be active; the result is wave-uniform (read from lane 63).
template<int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov_i32(int fill, int v)
{
    return __builtin_amdgcn_update_dpp(fill, v, CTRL, ROW_MASK, 0xf, false);
}
template<int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_mov_i64(long long fill, long long v)
{
    const unsigned lo = (unsigned) dpp_mov_i32<CTRL, ROW_MASK>((int) (unsigned) (unsigned long long) fill, (int) (unsigned) (unsigned long long) v);
    const unsigned hi = (unsigned) dpp_mov_i32<CTRL, ROW_MASK>((int) (unsigned) ((unsigned long long) fill >> 32),
                                                               (int) (unsigned) ((unsigned long long) v >> 32));
    return (long long) (((unsigned long long) hi << 32) | lo);
}
#define CC_DPP_REDUCE(T, MOV, v, fill, better)                   \
    {                                                            \
        T t_;                                                    \
        t_ = MOV<0x111, 0xf>(fill, v); v = better(t_, v) ? t_ : v; \
        t_ = MOV<0x112, 0xf>(fill, v); v = better(t_, v) ? t_ : v; \
        t_ = MOV<0x114, 0xf>(fill, v); v = better(t_, v) ? t_ : v; \
        t_ = MOV<0x118, 0xf>(fill, v); v = better(t_, v) ? t_ : v; \
        t_ = MOV<0x142, 0xa>(fill, v); v = better(t_, v) ? t_ : v; \
        t_ = MOV<0x143, 0xc>(fill, v); v = better(t_, v) ? t_ : v; \
    }
#define CC_LESS(a, b) ((a) < (b))
#define CC_GREATER(a, b) ((a) > (b))
__device__ __forceinline__ long long lane63_i64(long long v)
{
    const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (unsigned long long) v, 63);
    const unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) ((unsigned long long) v >> 32), 63);
    return (long long) (((unsigned long long) hi << 32) | lo);
}
__device__ __forceinline__ long long wave_min_i64(long long v)
{
    const long long fill = 0x7fffffffffffffffll;
    CC_DPP_REDUCE(long long, dpp_mov_i64, v, fill, CC_LESS)
    return lane63_i64(v);
}
__device__ __forceinline__ long long wave_max_i64(long long v)
{
    const long long fill = (long long) 0x8000000000000000ull;
    CC_DPP_REDUCE(long long, dpp_mov_i64, v, fill, CC_GREATER)
    return lane63_i64(v);
}
__device__ __forceinline__ int wave_min_i32(int v)
{
    const int fill = 0x7fffffff;
    CC_DPP_REDUCE(int, dpp_mov_i32, v, fill, CC_LESS)
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double wave_min_f64(double v)
{
    // (no NaNs reach this: azimuths and finished_at values)
    const long long fill = 0x7ff0000000000000ll; // +inf
    long long b = __double_as_longlong(v);
#define CC_LESS_F64(a, b) (__longlong_as_double(a) < __longlong_as_double(b))
    CC_DPP_REDUCE(long long, dpp_mov_i64, b, fill, CC_LESS_F64)
#undef CC_LESS_F64
    return __longlong_as_double(lane63_i64(b));
}
// maximum of non-negative, non-NaN doubles given as their bit patterns (finished_at values): v_max_f64 per step instead of a 64-bit
// compare and two selects
__device__ __forceinline__ unsigned long long wave_max_f64_bits(unsigned long long bits)
{
    long long b = (long long) bits;
#define CC_MAXF64(t, v) (__longlong_as_double(t) > __longlong_as_double(v))
    {
        long long t_;
        t_ = dpp_mov_i64<0x111, 0xf>(0ll, b); b = __double_as_longlong(__builtin_fmax(__longlong_as_double(t_), __longlong_as_double(b)));
        t_ = dpp_mov_i64<0x112, 0xf>(0ll, b); b = __double_as_longlong(__builtin_fmax(__longlong_as_double(t_), __longlong_as_double(b)));
        t_ = dpp_mov_i64<0x114, 0xf>(0ll, b); b = __double_as_longlong(__builtin_fmax(__longlong_as_double(t_), __longlong_as_double(b)));
        t_ = dpp_mov_i64<0x118, 0xf>(0ll, b); b = __double_as_longlong(__builtin_fmax(__longlong_as_double(t_), __longlong_as_double(b)));
        t_ = dpp_mov_i64<0x142, 0xa>(0ll, b); b = __double_as_longlong(__builtin_fmax(__longlong_as_double(t_), __longlong_as_double(b)));
        t_ = dpp_mov_i64<0x143, 0xc>(0ll, b); b = __double_as_longlong(__builtin_fmax(__longlong_as_double(t_), __longlong_as_double(b)));
    }
#undef CC_MAXF64
    return (unsigned long long) lane63_i64(b);
}
// Single-wavefront workgroups: LDS operations of one wave execute in issue order, so ordering LDS writes before LDS reads
// of other lanes needs neither s_barrier nor a vmcnt drain (which __syncthreads() implies and which would expose the
// latency of every global prefetch in flight). This is a compiler barrier plus a wait for outstanding LDS operations only.
__device__ __forceinline__ void wave_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Tell the compiler that a value is wave-uniform (it then lives in SGPRs and drives scalar branches instead of exec-masked
// "divergent" control flow). Only call with values that really are equal in all active lanes.
// Single-wave blocks: the LDS executes one wave's DS instructions in issue order, so a ds_write followed by a ds_read of the same
// word is ordered by the hardware even across lanes. Only the compiler has to be kept from moving LDS accesses across the point —
// no s_waitcnt (which would stall ~100 cycles per use for the stores to drain).
__device__ __forceinline__ void wave_lds_fence()
{
    asm volatile("" ::: "memory");
}

// Re-read / publish an LDS word that another lane or wave may change. Relaxed workgroup-scope atomics rather than volatile:
// the compiler leaves volatile accesses in the generic address space (flat_load ... sc0 sc1 followed by s_waitcnt vmcnt(0),
// which also drains every outstanding global load and store of the wave), while these become plain ds_read / ds_write.
template<class T>
__device__ __forceinline__ T lds_ld(const T* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template<class T>
__device__ __forceinline__ void lds_st(T* p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// element `ci` of a per-stream plane through a 32-bit BYTE offset: a wave-uniform base pointer plus a zero-extended 32-bit lane offset is the
// addressing mode global loads / stores have (saddr + voffset) — with a 64-bit index every access pays two or three instructions of address
// arithmetic. A stream's planes stay far below 4 GB (ring_cols * rows * 16 B).
template<class T>
__device__ __forceinline__ T& at32(T* base, const unsigned ci)
{
    return *(T*) ((char*) base + ci * (unsigned) sizeof(T));
}
__device__ __forceinline__ int uniform_i32(int v)
{
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ long long uniform_i64(long long v)
{
    const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (unsigned long long) v);
    const unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) ((unsigned long long) v >> 32));
    return (long long) (((unsigned long long) hi << 32) | lo);
}
__device__ __forceinline__ double uniform_f64(double v)
{
    return __longlong_as_double(uniform_i64(__double_as_longlong(v)));
}

// the value lane u holds, as a wave-uniform scalar (v_readlane)
__device__ __forceinline__ long long lane_i64(long long v, int u)
{
    const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (unsigned long long) v, u);
    const unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) ((unsigned long long) v >> 32), u);
    return (long long) (((unsigned long long) hi << 32) | lo);
}

// helpers: lane-indexed per-column scalars of a group (lane u holds column u's value)
__device__ __forceinline__ int lane_i32(int v, int u)
{
    return __builtin_amdgcn_readlane(v, u);
}
__device__ __forceinline__ double lane_f64(double v, int u)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (unsigned long long) b, u);
    const unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) ((unsigned long long) b >> 32), u);
    return __longlong_as_double((long long) (((unsigned long long) hi << 32) | lo));
}

__device__ __forceinline__ unsigned long long lanes_below()
{
    return (1ull << lane_id()) - 1ull;
}

// agent-scope relaxed accesses (bypass the CU's L1): used for every word that is also touched by atomics
template<class T>
__device__ __forceinline__ T ld_agent(const T* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void raise_error(StreamState* st, int code, long long a, long long b)
{
    if (atomicCAS(&st->error, 0, code) == 0)
    {
        st->error_a = a;
        st->error_b = b;
    }
}

// Point::associated_trees (cc.cpp:693-694) is an unordered union here; with Geometry::mirror_fields every link that is made is also
// logged as a pair of root cells, so that the host mirror can walk the tree graph in the reference's order (cc.cpp:851-910).
__device__ __forceinline__ void log_link(const Geometry& g, StreamState* st, int2* log, int cell_a, int cell_b)
{
    if (!g.mirror_fields)
        return;
    const int k = atomicAdd(&st->n_links, 1);
    if (k < g.link_capacity)
        log[k] = make_int2(cell_a, cell_b);
}

__device__ __forceinline__ uint16_t sat_u16(int v)
{
    return (uint16_t) (v > 65535 ? 65535 : v);
}

// ---- continuous azimuth angle of a cell (cc.cpp:184-186): 2 pi * rotation index + increasing azimuth angle, in double. The cell keeps the
// f32 increasing azimuth (Planes::incaz); the rotation index is that of the cell's global column, or one less when the sign bit is set.
// Same expression as the reference, so the same bits — at half the bytes of a stored double.
__device__ __forceinline__ float pack_incaz(const float inc_az, const bool previous_rotation)
{
    return previous_rotation ? __uint_as_float(__float_as_uint(inc_az) | 0x80000000u) : inc_az; // (inc_az >= +0: atan2f + pi)
}
struct CazBase
{
    double b0, b1; // 2 pi * rotation of the column, 2 pi * (rotation - 1)
};
__device__ __forceinline__ CazBase caz_base_of_rotation(const long long rot)
{
    CazBase b;
    b.b0 = CC_2PI_D * (double) rot;
    b.b1 = CC_2PI_D * (double) (rot - 1);
    return b;
}
__device__ __forceinline__ CazBase caz_base_of_column(const long long gc, const int num_columns)
{
    return caz_base_of_rotation(gc / num_columns); // (64-bit division: once per column, never per cell)
}
__device__ __forceinline__ double cell_caz(const CazBase& b, const float packed)
{
    const unsigned u = __float_as_uint(packed);
    return ((u >> 31) ? b.b1 : b.b0) + (double) __uint_as_float(u & 0x7fffffffu);
}
// a cell without a return: middle of its column (cc.cpp:371-372)
__device__ __forceinline__ double empty_cell_caz(const long long gc, const float az_width)
{
    return ((double) gc + 0.5) * (double) az_width;
}

// The smallest continuous azimuth over the cells of a column (Planes::colminaz) without a double per cell: cell_caz is monotone in the packed f32
// inside each of its two classes (this rotation / the previous one), so the minimum over a class is cell_caz of the class's smallest f32 —
// two 32-bit wave reductions and three f64 operations per column instead of an f64 add and compare per cell and a 64-bit reduction.
// kpos / kneg: this lane's smallest |packed| bits per class (0x7fffffff: none); any_empty: some cell of the column has no return.
__device__ __forceinline__ double column_min_caz(const CazBase& b, int kpos, int kneg, const bool any_empty, const long long gc, const float az_width)
{
    kpos = wave_min_i32(kpos);
    kneg = wave_min_i32(kneg);
    double m = 1.7976931348623157e308;
    if (kneg != 0x7fffffff)
        m = b.b1 + (double) __int_as_float(kneg);
    if (kpos != 0x7fffffff)
    {
        const double c = b.b0 + (double) __int_as_float(kpos);
        m = c < m ? c : m;
    }
    if (__any(any_empty))
    {
        const double c = empty_cell_caz(gc, az_width);
        m = c < m ? c : m;
    }
    return m;
}
// the (class, key) of one cell for column_min_caz
__device__ __forceinline__ void caz_key(const float packed, int& kpos, int& kneg)
{
    const unsigned u = __float_as_uint(packed);
    const int k = (int) (u & 0x7fffffffu);
    if (u >> 31)
        kneg = k < kneg ? k : kneg;
    else
        kpos = k < kpos ? k : kpos;
}

// ---- which pass over the ring a cell belongs to (Planes::gtag). The reference keeps the 64-bit global column index in every cell
// (Point::global_column_index, cleared to -1: cc.cpp:1110-1119) and compares it with the column being segmented (cc.cpp:320-345). A cell
// of ring column lc can only ever hold a global column lc + pass * ring_cols, so the pass index says the same in two bytes:
// 0 = cleared, else 0x8000 | (pass mod 2^15). A stale cell is met (and reported) on the very next pass, long before a tag could repeat.
constexpr uint16_t CELL_CLEARED = 0;
__device__ __forceinline__ uint16_t cell_tag(const long long pass)
{
    return (uint16_t) (0x8000u | ((unsigned) pass & 0x7fffu));
}

constexpr int IP_MAXF = 4608; // firings of a batch k_insert_par can take (LDS tables of its block scan)

// Pointers of one stream (planes offset to the stream's first cell / column / pool slot).
struct SP
{
    float *dist, *incl, *tabc;
    float* incaz;
    uint16_t* gtag;
    uint32_t* src;
    uint8_t *inten, *ground, *debug, *ignored;
    int32_t* trig;
    int64_t* colg;
    double* colminaz;
    int32_t* root;
    uint32_t* id;
    double* t_fin;
    uint32_t *t_width, *t_pts, *t_cid;
    int32_t *t_uf, *t_pos;
    uint8_t* t_finished;
    int32_t *ulist, *ucomp;
    unsigned long long* agg_fin;
    long long *agg_min, *agg_max;
    uint32_t *agg_pts, *agg_cid;
    int32_t* agg_first;
    uint8_t* agg_flag;
    float* curtab;
    unsigned long long* tab_acc;
    int32_t* par_off;
    cc_event* events;
    int16_t* sc_parent;
    int16_t* sc_term;
    double* col_newfin;
    int32_t* col_info;
    uint16_t* col_act;
    unsigned* pk_meta;
    double* pk_fin;
    unsigned long long* pk_lk;
    uint8_t* sc_nlinks;
    unsigned long long* sc_links;
    double* sc_fin;
    float *sg_x2, *sg_uz, *sg_w;
    uint8_t* sg_flags;
    float4* sc_rec;
    uint16_t* sc_visits;
    int2* link_log;
};

__device__ __forceinline__ SP stream_ptrs(const Planes& P, const Geometry& g, int s)
{
    SP p;
    const size_t co = (size_t) s * (size_t) g.cells;
    const size_t lo = (size_t) s * (size_t) g.ring_cols;
    const size_t to = (size_t) s * (size_t) g.tree_capacity;
    p.dist = P.dist + co;
    p.incl = P.incl + co;
    p.incaz = P.incaz + co;
    p.gtag = P.gtag + co;
    p.src = P.src + co;
    p.inten = P.inten + co;
    p.ground = P.ground + co;
    p.debug = P.debug + co;
    p.ignored = P.ignored + co;
    p.trig = P.trig + lo;
    p.colg = P.colg + lo;
    p.colminaz = P.colminaz + lo;
    p.root = P.root + co;
    p.id = P.id + co;
    p.t_fin = P.t_fin + co;
    p.t_width = P.t_width + co;
    p.t_pts = P.t_pts + co;
    p.t_cid = P.t_cid + co;
    p.t_uf = P.t_uf + co;
    p.t_pos = P.t_pos + co;
    p.t_finished = P.t_finished + co;
    p.ulist = P.ulist + to;
    p.ucomp = P.ucomp + to;
    p.agg_fin = P.agg_fin + to;
    p.agg_min = P.agg_min + to;
    p.agg_max = P.agg_max + to;
    p.agg_pts = P.agg_pts + to;
    p.agg_cid = P.agg_cid + to;
    p.agg_first = P.agg_first + to;
    p.agg_flag = P.agg_flag + to;
    p.curtab = P.curtab + (size_t) s * g.num_rows;
    p.par_off = P.par_off + (size_t) s * IP_MAXF;
    p.tab_acc = P.tab_acc + (size_t) s * (size_t) g.tab_tiles * g.num_rows;
    p.tabc = P.tabc + (size_t) s * (size_t) g.tab_tiles * g.num_rows;
    p.events = P.events + (size_t) s * g.event_capacity;
    p.sc_parent = P.sc_parent + co;
    p.sc_term = P.sc_term + co;
    p.col_newfin = P.col_newfin + lo;
    p.col_info = P.col_info + lo;
    p.col_act = P.col_act + lo;
    p.pk_meta = P.pk_meta + co;
    p.pk_fin = P.pk_fin + co;
    p.pk_lk = P.pk_lk + co;
    p.sc_nlinks = P.sc_nlinks + co;
    p.sc_links = P.sc_links + co;
    p.sc_fin = P.sc_fin + co;
    p.sg_x2 = P.sg_x2 + co;
    p.sg_uz = P.sg_uz + co;
    p.sg_w = P.sg_w + co;
    p.sg_flags = P.sg_flags + co;
    p.sc_rec = P.sc_rec + co;
    p.sc_visits = P.sc_visits + co;
    p.link_log = P.link_log + (size_t) s * (size_t) g.link_capacity;
    return p;
}

// =====================================================================================================
// ground-point segmentation — continuous_clustering.cpp:294-624, split into k_seg_pre (per cell) and k_seg_scan (per column)
// =====================================================================================================
__device__ __forceinline__ float len2(float a, float b)
{
    return ccm::sqrt_rn(a * a + b * b);
}

constexpr int EGO_STRIDE = 16; // doubles per firing in k_ego's output: {R 3x3, t, skip_r2, -}

// ---- the per-cell part of the segmentation of ONE column (everything of cc.cpp:306-403, 567-603 that does not depend on other columns or on
// the rows below), lanes = rows, cells in registers. Shared by k_seg_pre (cells from the ring) and k_insert_par (cells it has just computed).
//   x, y, z, dist, incl : the cell (odom frame; dist = incl = NaN without a return), inten its intensity
//   sp*                 : sgps_sensor_position of the column's job (the finishing firing's pose, cc.cpp:111-113, 291)
//   E                   : that firing's k_ego record (wave-uniform pointer: scalar loads)
// Staging for k_seg_scan: x2, uz (the point in the azimuth plane of the job's sensor position), flags (SG_*), and ONE more float w:
//   cell with a return, inclination step to the row below valid  w = that step (the column's own entry of the table, cc.cpp:353-357: k_seg_scan
//                                                                  takes the last valid one along the columns), cc.cpp:597-603 decided here
//   cell with a return, step not valid (SG_PENDING)              w = distance (k_seg_scan evaluates cc.cpp:597-603 once it knows the table)
//   cell without a return (SG_NAN)                               w = raw inclination of the row below (where the supplement chain of
//                                                                  cc.cpp:364-369 starts when that row has a return)
template<int RPL>
__device__ __forceinline__ void seg_pre_cells(const cc_config& cfg, const int R, const int lane, const float (&cx)[RPL], const float (&cy)[RPL],
                                              const float (&cz)[RPL], const float (&dist)[RPL], const float (&incl)[RPL], const uint8_t (&inten)[RPL],
                                              const float spx, const float spy, const float spz, const double* __restrict__ E, float (&x2)[RPL],
                                              float (&uz)[RPL], float (&w)[RPL], int (&flags)[RPL])
{
    // raw inclination of the row below (0 below the last row, cc.cpp:312)
    float below[RPL];
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        const float nxt0 = (k + 1 < RPL) ? __shfl(incl[(k + 1 < RPL) ? k + 1 : k], 0, 64) : 0.f;
        const float dn = __shfl_down(incl[k], 1, 64);
        below[k] = lane == 63 ? nxt0 : dn;
        if (k * 64 + lane + 1 >= R)
            below[k] = 0.f;
    }
    const float skip_r2 = (float) E[12];
    bool close = false, need_exact = false;
    bool incl_ignore[RPL];
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        const int row = k * 64 + lane;
        flags[k] = SG_NAN;
        x2[k] = uz[k] = 0.f;
        w[k] = below[k];
        incl_ignore[k] = false;
        if (row >= R)
            continue;
        const bool isnan_ = dist[k] != dist[k];
        if (isnan_)
            continue;
        int f = 0;
        if (cfg.fog_filtering_enabled && inten[k] < (uint8_t) cfg.fog_filtering_intensity_below && dist[k] < cfg.fog_filtering_distance_below &&
            incl[k] > cfg.fog_filtering_inclination_above)
            f |= SG_FOG;
        const float ux = cx[k] - spx, uy = cy[k] - spy;
        uz[k] = cz[k] - spz;
        x2[k] = len2(ux, uy);
        const float r2 = x2[k] * x2[k] + uz[k] * uz[k];
        if (!(r2 > skip_r2))
        {
            f |= SG_EGO; // provisional: "needs the transform"
            close = true;
        }
        if (dist[k] < cfg.max_distance) // (cc.cpp:590: distance < 1. * max_distance in double — both convert exactly, the same comparison)
            f |= SG_TOO_CLOSE;
        const float diff = incl[k] - below[k];
        if (diff != diff)
        {
            f |= SG_PENDING;
            w[k] = dist[k];
        }
        else
        {
            w[k] = diff;
            // cc.cpp:597-603: atan2f(max_distance, distance) < inclination step to the next laser. The exact (glibc-identical) atan2f
            // costs ~100 instructions per wave, and the test can only hold beyond ~100 m: a rigorous filter first. With
            // x = max_distance / distance >= 1.01 t (0 <= t < 0.05): atan(x) >= x - x^3/3 >= 1.006 t for x <= 0.1, atan(x) > 0.099 > t
            // otherwise, and atan2f is within an ulp of atan — so the test is false without evaluating it.
            if (cfg.ignore_points_with_too_big_inclination_angle_diff && row < (R - 1))
            {
                const bool surely_false = cfg.max_distance > 0.f && diff >= 0.f && diff < 0.05f && cfg.max_distance >= 1.01f * dist[k] * diff;
                incl_ignore[k] = !surely_false; // provisional: "needs the exact evaluation"
                need_exact |= !surely_false;
            }
        }
        flags[k] = f;
    }
    if (__any(need_exact))
    {
#pragma unroll
        for (int k = 0; k < RPL; k++)
            if (incl_ignore[k])
                incl_ignore[k] = ccm::atan2f_exact(cfg.max_distance, dist[k]) < w[k];
    }
#pragma unroll
    for (int k = 0; k < RPL; k++)
        if (incl_ignore[k])
            flags[k] |= SG_INCL_IGNORE;
    if (__any(close))
    {
        // ego_robot_frame_from_odom_frame * point (cc.cpp:390-403), Eigen's evaluation order
        double er[9], et[3];
        for (int i = 0; i < 9; i++)
            er[i] = E[i];
        for (int i = 0; i < 3; i++)
            et[i] = E[9 + i];
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            if (!(flags[k] & SG_EGO) || (flags[k] & SG_NAN))
                continue;
            const double dx = cx[k], dy = cy[k], dz = cz[k];
            const double ex = ((er[0] * dx + er[1] * dy) + er[2] * dz) + et[0];
            const double ey = ((er[3] * dx + er[4] * dy) + er[5] * dz) + et[1];
            const double ez = ((er[6] * dx + er[7] * dy) + er[8] * dz) + et[2];
            const bool in_box = ex < cfg.length_ref_to_front_end_ && ex > cfg.length_ref_to_rear_end_ && ey < cfg.width_ref_to_left_mirror_ &&
                                ey > cfg.width_ref_to_right_mirror_ && ez < cfg.height_ref_to_maximum_ && ez > cfg.height_ref_to_ground_;
            if (!in_box)
                flags[k] &= ~SG_EGO;
        }
    }
}


// =====================================================================================================
// k_prep — the per-point part of insertFiringIntoRangeImage (cc.cpp:127-151, 189, 224-232): rigid transform, range,
// azimuth -> column within the rotation, inclination. Independent per point, so it runs over all points of the batch
// in parallel; the serial kernel below only decides where each point lands. grid = points / 256, block = 256.
// =====================================================================================================
// The per-point arithmetic, shared by k_prep and k_insert_par so that both produce the same bits.
struct PreppedPoint
{
    float x, y, z, dist, incl, incaz;
    int cir; // column within the rotation, PP_SKIP for a NaN return
};

__device__ __forceinline__ PreppedPoint prep_point(const float fx, const float fy, const float fz, const double* __restrict__ T, const bool clockwise,
                                                   const float az_width)
{
    PreppedPoint o;
    o.x = o.y = o.z = o.dist = o.incl = o.incaz = 0.f;
    o.cir = PP_SKIP; // std::isnan(p.x()) cc.cpp:131
    if (fx != fx)
        return o;
    const double px = fx, py = fy, pz = fz;
    const double tx = T[3], ty = T[7], tz = T[11];
    const double ox = ((T[0] * px + T[1] * py) + T[2] * pz) + tx;
    const double oy = ((T[4] * px + T[5] * py) + T[6] * pz) + ty;
    const double oz = ((T[8] * px + T[9] * py) + T[10] * pz) + tz;
    const double rx = ox - tx, ry = oy - ty, rz = oz - tz;
    const float az = ccm::atan2f_exact(fy, fx);
    const float inc_az = clockwise ? -az + CC_PI_F : az + CC_PI_F;
    const float dist = (float) __builtin_sqrt((rx * rx + ry * ry) + rz * rz);
    o.x = (float) ox;
    o.y = (float) oy;
    o.z = (float) oz;
    o.dist = dist;
    o.incl = ccm::asinf_exact((float) rz / dist);
    o.incaz = inc_az;
    o.cir = f2i_x86(inc_az / az_width);
    return o;
}

// grid = (points of one stream's sub-batch / PREP_POINTS_PER_BLOCK, streams). The caller's buffers hold n_total firings per stream; this launch
// prepares firings [f0, f0 + m) of every stream into the compact staging planes (index [stream][m][row]). Firings that k_insert_par
// has already inserted (below the stream's cursor) are skipped: nobody reads their staging cells.
constexpr int PREP_POINTS_PER_BLOCK = 4096; // 16 rounds of 256 threads: few, fat blocks — when k_insert_par has taken the whole batch every
                                             // block leaves after one test, and 9 k blocks do that faster than 140 k
__global__ __launch_bounds__(256) void k_prep(Geometry g, cc_config cfg, Planes P, const float* __restrict__ xyz,
                                             const double* __restrict__ poses, long long m, long long n_total, long long f0,
                                             const StreamState* __restrict__ states, int first_stream)
{
    const int R = g.num_rows;
    const long long sl = blockIdx.y;
    const long long cursor = states ? states[first_stream + sl].cursor : 0;
    const long long block_first = (long long) blockIdx.x * PREP_POINTS_PER_BLOCK;
    const long long total = m * R;
    if (block_first >= total || (block_first + PREP_POINTS_PER_BLOCK - 1) / R < cursor)
        return; // every firing of this block has been inserted already
    for (long long local = block_first + threadIdx.x; local < block_first + PREP_POINTS_PER_BLOCK && local < total; local += 256)
    {
        if (local / R < cursor)
            continue;
        const long long src = (sl * n_total + f0) * R + local; // index into the caller's [stream][n_total][row] buffers
        const long long firing = src / R;                      // [stream][firing] flattened
        const long long i = sl * m * R + local;                // index into the staging planes
        const PreppedPoint q = prep_point(xyz[src * 3 + 0], xyz[src * 3 + 1], xyz[src * 3 + 2], poses + firing * 12, cfg.sensor_is_clockwise != 0, g.az_width);
        P.pp_cir[i] = q.cir;
        if (q.cir == PP_SKIP)
            continue;
        P.pp_x[i] = q.x;
        P.pp_y[i] = q.y;
        P.pp_z[i] = q.z;
        P.pp_dist[i] = q.dist;
        P.pp_incl[i] = q.incl;
        P.pp_incaz[i] = q.incaz;
    }
}

// =====================================================================================================
// k_insert2 — the serial part of insertFiringIntoRangeImage (cc.cpp:152-292): global column of every return relative to the
// previous rearmost laser, cell collision rule, rearmost / foremost tracking, emission of finished columns. One wavefront
// per stream, lanes = rows; the `distance` plane of the INS_WIN columns around the insertion front lives in LDS so that the
// occupancy tests never wait for HBM.
// =====================================================================================================
#ifndef CC_INS_RING
#define CC_INS_RING 8
#endif
constexpr int INS_RING = CC_INS_RING;  // firings staged in LDS ahead of the consumer wave

// columns of the `distance` plane kept in LDS: INS_WIN for sensors whose firing spans a few columns, twice that for sensors with
// two rows per lane (VLS-128-style firings span ~60 columns)
__host__ __device__ constexpr int ins_win_cols(int rpl)
{
    return rpl == 1 ? INS_WIN : 2 * INS_WIN;
}

__host__ inline size_t insert2_lds_bytes(int R)
{
    // distance window + ring of staged firings (7 float/int planes + intensity) + 3 sync words
    const int rpl = (R + WAVE - 1) / WAVE;
    return (size_t) ins_win_cols(rpl) * R * 4 + (size_t) INS_RING * R * (7 * 4 + 4) + 64;
}

// block = 128: wavefront 0 is the consumer (the serial algorithm), wavefront 1 the loader that streams the staged points
// of the coming firings from HBM into an LDS ring, so that the consumer never waits for a global load.
// (a device function: k_insert2 is its kernel; k_small_front runs it behind the preparation of a small call, in the same block)
// NOWIN (k_small_front: a call of a few firings, where filling the LDS window of 64 columns — four dependent rounds of global loads — costs more
// than the call's handful of cells): no distance window, the occupancy tests read the global plane; results are the same by construction (the
// window is a cache of that plane: `res` selects between the two copies everywhere)
template<int RPL, bool NOWIN = false>
__device__ __forceinline__ void insert2_body(const Geometry& g, const cc_config& cfg, const Planes& P, StreamState* states, int first_stream, int slot,
                                             const uint8_t* __restrict__ inten, long long n, int* remaining, long long n_total, long long fbase,
                                             const int sl)
{
    const int s = first_stream + sl;
    const int lane = lane_id();
    const int wave = uniform_i32((int) (threadIdx.x >> 6));
    StreamState* st = &states[s];
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, NC = g.num_columns, RC = g.ring_cols;
    constexpr int WINC = ins_win_cols(RPL);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* w_dist = (float*) smem;                       // [WINC][R]
    float* r_x = w_dist + WINC * R;                   // [INS_RING][R] each
    float* r_y = r_x + INS_RING * R;
    float* r_z = r_y + INS_RING * R;
    float* r_d = r_z + INS_RING * R;
    float* r_i = r_d + INS_RING * R;
    float* r_a = r_i + INS_RING * R;
    int* r_c = (int*) (r_a + INS_RING * R);
    int* r_t = r_c + INS_RING * R;                       // intensity (one int per cell keeps the stores conflict-free)
    long long* v_ready = (long long*) (r_t + INS_RING * R); // firings [.., v_ready) are staged
    long long* v_done = v_ready + 1;                     // firings [.., v_done) have been consumed
    long long* v_stop = v_ready + 2;                     // consumer stopped early at this firing (or -1)

    const long long cursor0 = st->cursor;
    const size_t pbase = (size_t) sl * (size_t) n * R;
    if (threadIdx.x == 0)
    {
        lds_st(v_ready, cursor0);
        lds_st(v_done, cursor0);
        lds_st(v_stop, -1ll);
    }
    __syncthreads();

    if (wave == 1)
    {
        // ------------------------------------------------------------------ loader
        const uint8_t* si = inten + ((size_t) sl * (size_t) n_total + (size_t) fbase) * R; // caller's [stream][n_total][row] buffer
        const float *qx = P.pp_x + pbase, *qy = P.pp_y + pbase, *qz = P.pp_z + pbase, *qd = P.pp_dist + pbase, *qi = P.pp_incl + pbase,
                    *qa = P.pp_incaz + pbase;
        const int32_t* qc = P.pp_cir + pbase;
        constexpr int U = 4; // firings in flight per round
        for (long long f0 = cursor0; f0 < n; f0 += U)
        {
            float x[U][RPL], y[U][RPL], z[U][RPL], d[U][RPL], ii[U][RPL], a[U][RPL];
            int c[U][RPL], t[U][RPL];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    c[u][k] = PP_SKIP;
                    x[u][k] = y[u][k] = z[u][k] = d[u][k] = ii[u][k] = a[u][k] = 0.f;
                    t[u][k] = 0;
                    if (row < R && f0 + u < n)
                    {
                        const size_t pi = (size_t) (f0 + u) * R + row;
                        c[u][k] = qc[pi];
                        x[u][k] = qx[pi];
                        y[u][k] = qy[pi];
                        z[u][k] = qz[pi];
                        d[u][k] = qd[pi];
                        ii[u][k] = qi[pi];
                        a[u][k] = qa[pi];
                        t[u][k] = si[pi];
                    }
                }
            // wait until the ring has room for these U firings (or the consumer stopped)
            while (lds_ld(v_done) + INS_RING < f0 + U && lds_ld(v_stop) < 0)
                __builtin_amdgcn_s_sleep(2);
            if (lds_ld(v_stop) >= 0)
                break;
#pragma unroll
            for (int u = 0; u < U; u++)
            {
                const int slot = (int) ((f0 + u) % INS_RING);
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    if (row < R)
                    {
                        const int o = slot * R + row;
                        r_c[o] = c[u][k];
                        r_x[o] = x[u][k];
                        r_y[o] = y[u][k];
                        r_z[o] = z[u][k];
                        r_d[o] = d[u][k];
                        r_i[o] = ii[u][k];
                        r_a[o] = a[u][k];
                        r_t[o] = t[u][k];
                    }
                }
            }
            wave_lds_sync();
            if (lane == 0)
                lds_st(v_ready, (long long) (f0 + U < n ? f0 + U : n));
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer
    __builtin_amdgcn_s_setprio(3); // latency-critical serial chain: win issue arbitration against co-resident throughput kernels
    long long prev_rear = st->prev_rearmost, prev_fore = st->prev_foremost, first_unf = st->first_unfinished;
    long long ring_start = st->ring_start, ring_end = st->ring_end, first_unpub = st->first_unpublished;
    int reset_required = st->reset_required;
    const long long seq0 = (long long) st->firings_consumed;
    long long seg_begin = first_unf;
    long long limit_base = first_unf;
    if (st->pre_seg_begin > 0)
    {
        // k_insert_par consumed the head of this batch: the batch's column range and its emission limit start where it started
        seg_begin = st->pre_seg_begin;
        limit_base = st->pre_seg_begin;
    }
    unsigned long long negative_cols = 0;
    bool ring_init = false;

    // deferred clearColumns (cc.cpp:1094-1145) for what earlier calls released
    long long clear_done = st->clear_done;
    if (clear_done >= 0)
    {
        const long long clear_to = ring_start < st->clear_allowed ? ring_start : st->clear_allowed;
        for (; clear_done < clear_to; clear_done++)
        {
            const int clc = (int) (clear_done % RC);
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (row < R)
                {
                    const size_t ci = (size_t) clc * R + row;
                    p.dist[ci] = __builtin_nanf("");
                    p.incl[ci] = __builtin_nanf("");
                    p.gtag[ci] = CELL_CLEARED;
                }
            }
        }
    }

    // window = global columns [wbase, wbase + WINC), column gcx at LDS column gcx % WINC
    long long wbase = -1;
    auto window_fill = [&](long long from, long long to) // load columns [from, to) from the global distance plane
    {
        int lcx = (int) (from % RC);
        constexpr int B = 16; // columns in flight
        for (long long g0 = from; g0 < to; g0 += B)
        {
            float v[B][RPL];
            int lcs = lcx;
#pragma unroll
            for (int u = 0; u < B; u++)
            {
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    v[u][k] = 0.f;
                    if (row < R && g0 + u < to)
                        v[u][k] = p.dist[(size_t) lcs * R + row];
                }
                lcs = lcs + 1 == RC ? 0 : lcs + 1;
            }
#pragma unroll
            for (int u = 0; u < B; u++)
            {
                const int wc = (int) ((g0 + u) & (WINC - 1));
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    if (row < R && g0 + u < to)
                        w_dist[wc * R + row] = v[u][k];
                }
            }
            lcx = lcs;
        }
    };
    auto window_seek = [&](long long need_lo, long long need_hi) // make [need_lo, need_hi] resident if it fits
    {
        long long nb = need_lo - 24;
        if (nb < 0)
            nb = 0;
        if (wbase < 0 || nb >= wbase + WINC || nb < wbase)
        {
            wbase = nb;
            window_fill(wbase, wbase + WINC);
        }
        else if (need_hi >= wbase + WINC)
        {
            window_fill(wbase + WINC, nb + WINC);
            wbase = nb;
        }
        wave_lds_sync();
    };

    // 64-bit divisions by run-time divisors cost hundreds of cycles each: keep rotation index, column within the rotation
    // and ring column of the previous rearmost laser incrementally
    long long prev_rot = prev_rear / NC;
    int prev_cir = (int) (prev_rear - prev_rot * NC);
    int rear_lc = (int) (prev_rear % RC);
    long long rear_pass = prev_rear / RC; // pass over the ring the previous rearmost laser is in (cell_tag)
    long long tracked_rear = prev_rear;
#ifdef CC_PROFILE_SECTIONS
    unsigned long long isec[6] = {0, 0, 0, 0, 0, 0};
#define CC_ISEC(i) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); isec[i] += _n - ins_work_mark; ins_work_mark = _n; }
    unsigned long long ins_wait = 0, ins_work = 0, ins_work_mark = 0;
    const unsigned long long ins_t0 = __builtin_amdgcn_s_memtime();
#endif
    long long f = cursor0;
    for (; f < n; f++)
    {
        // ---- tight loop over the common firing shape: every return in one and the same column (kitti_demo's pseudo firings,
        // kd.cpp:123-159), no rotation wrap relative to the previous rearmost laser, the column inside the LDS window, every
        // target cell empty, at most 64 columns to emit. Under exactly these conditions the general code below does the same;
        // here all state stays scalar and nothing of the generic bookkeeping is executed. A lone wavefront retires about one
        // instruction per 5-8 cycles, so the length of this loop body IS the insertion rate.
        if (tracked_rear != prev_rear)
        {
            const long long dlt = prev_rear - tracked_rear;
            if (dlt > 0 && dlt < NC)
            {
                prev_cir += (int) dlt;
                if (prev_cir >= NC)
                {
                    prev_cir -= NC;
                    prev_rot++;
                }
                rear_lc += (int) dlt;
                if (rear_lc >= RC)
                {
                    rear_lc -= RC;
                    rear_pass++;
                }
            }
            else
            {
                prev_rot = prev_rear / NC;
                prev_cir = (int) (prev_rear - prev_rot * NC);
                rear_lc = (int) (prev_rear % RC);
                rear_pass = prev_rear / RC;
            }
            tracked_rear = prev_rear;
        }
        if (ring_start != -1 && first_unf != -1 && prev_fore >= 0 && wbase >= 0)
        {
            const int half_ = NC / 2;
            long long ready_upto = f;
            while (f < n)
            {
                if (limit_base >= 0 && prev_rear - limit_base >= g.limit_columns)
                    break;
                if (ready_upto <= f)
                {
                    ready_upto = lds_ld(v_ready);
                    if (ready_upto <= f)
                    {
                        __builtin_amdgcn_s_sleep(1);
                        continue;
                    }
                    wave_lds_sync();
                }
                const int slot = (int) (f & (INS_RING - 1));
                int cirv[RPL];
                bool v[RPL];
                unsigned long long mv = 0;
                int c0 = 0;
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    cirv[k] = row < R ? r_c[slot * R + row] : PP_SKIP;
                    v[k] = cirv[k] != PP_SKIP;
                    const unsigned long long m = __ballot(v[k]);
                    if (mv == 0 && m != 0)
                        c0 = __builtin_amdgcn_readlane(cirv[k], (int) __ffsll((long long) m) - 1); // v_readlane: no LDS round trip
                    mv |= m;
                }
                if (mv == 0)
                    break;
                bool differs = false;
#pragma unroll
                for (int k = 0; k < RPL; k++)
                    differs |= v[k] && cirv[k] != c0;
                const int cdiff = c0 - prev_cir;
                const long long gc0 = prev_rot * NC + c0;
                if (__any(differs) || c0 < 0 || cdiff < -half_ || cdiff > half_ || gc0 < wbase || gc0 + 1 >= wbase + WINC ||
                    gc0 < first_unf || (gc0 > prev_rear && gc0 - first_unf > 64))
                    break;
                const int wcol = (int) (gc0 & (WINC - 1)) * R;
                bool occupied = false;
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    if (v[k])
                    {
                        const float cd = w_dist[wcol + row];
                        occupied |= !(cd != cd);
                    }
                }
                if (__any(occupied))
                    break;
                int lc = rear_lc + (int) (gc0 - prev_rear);
                long long pass = rear_pass;
                if (lc < 0)
                {
                    lc += RC;
                    pass--;
                }
                else if (lc >= RC)
                {
                    lc -= RC;
                    pass++;
                }
                const uint16_t tag0 = cell_tag(pass);
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    if (v[k])
                    {
                        const int so = slot * R + row;
                        const size_t ci = (size_t) lc * R + row;
                        const float d = r_d[so];
                        p.sc_rec[ci] = make_float4(r_x[so], r_y[so], r_z[so], r_i[so]);
                        p.inten[ci] = (uint8_t) r_t[so];
                        p.src[ci] = (uint32_t) (seq0 + (f - cursor0));
                        p.dist[ci] = d;
                        p.incl[ci] = r_i[so];
                        p.incaz[ci] = pack_incaz(r_a[so], c0 >= NC); // (rotation of the return: prev_rot, the column's unless c0 == NC)
                        p.gtag[ci] = tag0;
                        w_dist[wcol + row] = d;
                    }
                }
                // rear = fore = gc0 (cc.cpp:241-266)
                if (gc0 > prev_rear)
                {
                    const int dlt = (int) (gc0 - prev_rear);
                    prev_rear = gc0;
                    prev_cir += dlt;
                    if (prev_cir >= NC)
                    {
                        prev_cir -= NC;
                        prev_rot++;
                    }
                    rear_lc += dlt;
                    if (rear_lc >= RC)
                    {
                        rear_lc -= RC;
                        rear_pass++;
                    }
                    tracked_rear = prev_rear;
                }
                if (gc0 > prev_fore)
                    prev_fore = gc0;
                if (prev_fore > ring_end)
                    ring_end = prev_fore;
                // finished columns carry the pose of this firing (cc.cpp:289-291)
                if (first_unf < prev_rear)
                {
                    const int cnt = (int) (prev_rear - first_unf); // <= 64 by the entry condition
                    if (lane < cnt)
                    {
                        int tl = rear_lc - (cnt - lane);
                        if (tl < 0)
                            tl += RC;
                        p.trig[tl] = (int) f;
                    }
                    first_unf = prev_rear;
                }
                f++;
                if ((f & 3) == 0)
                {
                    wave_lds_sync();
                    if (lane == 0)
                        lds_st(v_done, (long long) f);
                }
            }
            wave_lds_sync();
            if (lane == 0)
                lds_st(v_done, (long long) f);
            if (f >= n || (limit_base >= 0 && prev_rear - limit_base >= g.limit_columns))
                break;
        }
        if (limit_base >= 0 && prev_rear - limit_base >= g.limit_columns)
            break;
#ifdef CC_PROFILE_SECTIONS
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
        while (lds_ld(v_ready) <= f)
            __builtin_amdgcn_s_sleep(1);
        wave_lds_sync();
#ifdef CC_PROFILE_SECTIONS
        const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
        ins_wait += t1_ - t0_;
        ins_work_mark = t1_;
#endif
        const int slot = (int) (f & (INS_RING - 1));
        if (tracked_rear != prev_rear)
        {
            const long long dlt = prev_rear - tracked_rear;
            if (dlt > 0 && dlt < NC)
            {
                prev_cir += (int) dlt;
                if (prev_cir >= NC)
                {
                    prev_cir -= NC;
                    prev_rot++;
                }
                rear_lc += (int) dlt;
                if (rear_lc >= RC)
                {
                    rear_lc -= RC;
                    rear_pass++;
                }
            }
            else
            {
                prev_rot = prev_rear / NC;
                prev_cir = (int) (prev_rear - prev_rot * NC);
                rear_lc = (int) (prev_rear % RC);
                rear_pass = prev_rear / RC;
            }
            tracked_rear = prev_rear;
        }
        int cir[RPL];
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            cir[k] = row < R ? r_c[slot * R + row] : PP_SKIP;
        }
        const int half = NC / 2;
        const long long rot_base = prev_rot * NC;
        // global column of every return (cc.cpp:152-175)
        long long gcv[RPL];
        int rot_off[RPL];
        bool have[RPL];
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            have[k] = cir[k] != PP_SKIP;
            gcv[k] = 0;
            rot_off[k] = 0;
            if (have[k])
            {
                long long gc = rot_base + cir[k];
                const int cdiff = cir[k] - prev_cir;
                if (cdiff < -half)
                {
                    gc += NC;
                    rot_off[k] = 1;
                }
                else if (prev_rear > 0 && cdiff > half)
                {
                    gc -= NC;
                    rot_off[k] = -1;
                }
                if (gc < 0)
                {
                    negative_cols++; // undefined behaviour in the reference (negative vector index); dropped here
                    have[k] = false;
                }
                gcv[k] = gc;
            }
        }
#ifdef CC_PROFILE_SECTIONS
        CC_ISEC(0)
#endif
        // wave-wide range of touched columns (DPP reductions: no LDS round trips)
        long long need_lo = 0x7fffffffffffffffll, need_hi = -1;
        {
            long long lo = 0x7fffffffffffffffll, hi = -1;
#pragma unroll
            for (int k = 0; k < RPL; k++)
                if (have[k])
                {
                    lo = gcv[k] < lo ? gcv[k] : lo;
                    hi = gcv[k] > hi ? gcv[k] : hi;
                }
            need_lo = wave_min_i64(lo);
            need_hi = wave_max_i64(hi);
        }
#ifdef CC_PROFILE_SECTIONS
        CC_ISEC(1)
#endif
        need_lo = uniform_i64(need_lo);
        need_hi = uniform_i64(need_hi);
        long long rear = -1, fore = -1;
        if (need_hi >= 0)
        {
            if (!NOWIN && (wbase < 0 || need_lo < wbase || need_hi + 1 >= wbase + WINC))
                window_seek(need_lo, need_hi + 1);
            long long l_rear = 0x7fffffffffffffffll, l_fore = -1;
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (!have[k])
                    continue;
                long long gc = gcv[k];
                // ring column: offset from the previous rearmost laser's ring column (|offset| < one rotation < RC)
                int lc = rear_lc + (int) (gc - prev_rear);
                long long pass = rear_pass; // pass over the ring of column gc (cell_tag)
                if (lc < 0)
                {
                    lc += RC;
                    pass--;
                }
                else if (lc >= RC)
                {
                    lc -= RC;
                    pass++;
                }
                const int so = slot * R + row;
                const float d = r_d[so];
                const bool res = !NOWIN && gc >= wbase && gc + 1 < wbase + WINC; // both candidate columns resident in LDS
                // (two separate loads, not a select between an LDS and a global address: that becomes a flat load, whose wait
                // drains every outstanding global store of the wave)
                // (the LDS read is unconditional and the global one an exception, so that the two are never merged into one flat
                // load)
                float cd = lds_ld(&w_dist[res ? (int) (gc & (WINC - 1)) * R + row : row]);
                if (__any(!res)) // (uniform test first: the exception stays a branch)
                {
                    if (!res)
                        cd = ld_agent(&p.dist[(size_t) lc * R + row]);
                }
                if (!(cd != cd) && !(d != d)) // cell occupied: try the next column (cc.cpp:188-202)
                {
                    float nd = lds_ld(&w_dist[res ? (int) ((gc + 1) & (WINC - 1)) * R + row : row]);
                    if (__any(!res))
                    {
                        if (!res)
                            nd = ld_agent(&p.dist[(size_t) (lc + 1 >= RC ? 0 : lc + 1) * R + row]);
                    }
                    if (nd != nd)
                    {
                        gc++;
                        pass += lc + 1 >= RC ? 1 : 0;
                        lc = lc + 1 >= RC ? 0 : lc + 1;
                        cd = nd;
                    }
                }
                if (!(cd != cd) && ((d != d) || d >= cd))
                    continue; // never overwrite a valid cell by NaN or a farther return (cc.cpp:204-206)
                const bool too_far_behind = first_unf >= 0 && gc < first_unf;
                if (!too_far_behind)
                {
                    const size_t ci = (size_t) lc * R + row;
#ifndef CC_EXP_NOSTORE
                    p.sc_rec[ci] = make_float4(r_x[so], r_y[so], r_z[so], r_i[so]);
                    p.inten[ci] = (uint8_t) r_t[so];
                    p.src[ci] = (uint32_t) (seq0 + (f - cursor0));
                    p.incl[ci] = r_i[so];
                    // rotation of the return = prev_rot + rot_off (cc.cpp:184-186) = that of its column gc = gcv (+ 1 if moved on), or one less
                    p.incaz[ci] = pack_incaz(r_a[so], cir[k] + (int) (gc - gcv[k]) >= NC);
                    p.gtag[ci] = cell_tag(pass);
#endif
                    p.dist[ci] = d;
                    if (!NOWIN && gc >= wbase && gc < wbase + WINC)
                        w_dist[(int) (gc & (WINC - 1)) * R + row] = d;
                }
                l_rear = gc < l_rear ? gc : l_rear;
                l_fore = gc > l_fore ? gc : l_fore;
            }
#ifdef CC_PROFILE_SECTIONS
            CC_ISEC(2)
#endif
            // rearmost / foremost over the lanes that reached the tracking code: values lie in [need_lo, need_hi + 1]
            {
                const int span = (int) (need_hi + 1 - need_lo);
                int o_lo = l_fore >= 0 ? (int) (l_rear - need_lo) : 0x7fffffff;
                int o_hi = l_fore >= 0 ? (int) (l_fore - need_lo) : -1;
                if (span <= 1)
                {
                    // KITTI-shaped firings: every return in one column (or its successor)
                    const unsigned long long lo0 = __ballot(o_lo == 0), hi1 = __ballot(o_hi == 1), any = __ballot(o_hi >= 0);
                    if (any)
                    {
                        rear = need_lo + (lo0 ? 0 : 1);
                        fore = need_lo + (hi1 ? 1 : 0);
                    }
                }
                else
                {
                    o_lo = wave_min_i32(o_lo);
                    int neg_hi = -o_hi;
                    neg_hi = wave_min_i32(neg_hi);
                    o_hi = -neg_hi;
                    if (o_hi >= 0)
                    {
                        rear = need_lo + o_lo;
                        fore = need_lo + o_hi;
                    }
                }
            }
        }
        rear = uniform_i64(rear);
        fore = uniform_i64(fore);
        wave_lds_sync();
        if (lane == 0)
            lds_st(v_done, (long long) (f + 1));
#ifdef CC_PROFILE_SECTIONS
        CC_ISEC(3)
#endif

        if (rear >= 0 && fore >= 0)
        {
            if ((fore - rear) > NC / 2)
            {
                reset_required = 1; // cc.cpp:252-261
                continue;
            }
            if (rear > prev_rear)
                prev_rear = rear;
            if (fore > prev_fore)
                prev_fore = fore;
        }
        if (prev_fore < 0)
            continue;
        if (ring_start == -1)
        {
            ring_start = prev_rear;
            first_unpub = prev_rear;
            clear_done = prev_rear;
            ring_init = true;
        }
        if (prev_fore > ring_end)
            ring_end = prev_fore;
        if (first_unf == -1)
        {
            first_unf = prev_rear;
            if (seg_begin < 0)
                seg_begin = first_unf;
            if (lane == 0)
                st->first_column = first_unf;
        }
        // finished columns carry the pose of this firing (cc.cpp:289-291)
        if (first_unf < prev_rear)
        {
            if (prev_rear - first_unf < RC)
            {
                // ring column of first_unf from the (already updated) rearmost column; tracked_* still describe the old one
                for (long long c = first_unf + lane; c < prev_rear; c += 64)
                {
                    int tl = rear_lc + (int) (c - tracked_rear);
                    if (tl < 0)
                        tl += RC;
                    else if (tl >= RC)
                        tl -= RC;
                    p.trig[tl] = (int) f;
                }
            }
            else
                for (long long c = first_unf + lane; c < prev_rear; c += 64)
                    p.trig[(int) (c % RC)] = (int) f;
            first_unf = prev_rear;
        }
    }
    if (lane == 0)
        lds_st(v_stop, (long long) f); // releases the loader if it is waiting for ring space
#ifdef CC_PROFILE_SECTIONS
    if (lane == 0)
    {
        st->dbg[0] += ins_wait;
        st->dbg[1] += isec[0];
        st->dbg[2] += isec[1];
        st->dbg[3] += isec[2];
        st->dbg[4] += isec[3];
        st->dbg[5] += __builtin_amdgcn_s_memtime() - ins_t0;
    }
#endif

    if (lane == 0)
    {
        st->prev_rearmost = prev_rear;
        st->prev_foremost = prev_fore;
        st->first_unfinished = first_unf;
        // ring_start / first_unpublished belong to the association chain (which may be running the previous batch right
        // now); the insertion kernel only gives them their initial value (cc.cpp:274-278)
        if (ring_init)
        {
            st->ring_start = ring_start;
            st->first_unpublished = first_unpub;
        }
        st->ring_end = ring_end;
        st->clear_done = clear_done;
        st->reset_required = reset_required;
        st->batch[slot].seg_begin = seg_begin;
        st->batch[slot].seg_end = seg_begin >= 0 ? first_unf : -1;
        st->batch[slot].acp_next = seg_begin;
        st->batch[slot].pub_begin = -1;
        st->batch[slot].pub_end = -1;
        if (cursor0 < n)
            st->batch[slot].fused = 0; // (nothing left for this kernel: the batch is k_insert_par's, and so is the flag)
        st->cursor = f;
        st->pre_seg_begin = 0;
        st->firings_consumed = (unsigned long long) (seq0 + (f - cursor0));
        if (f < n)
            atomicAdd(remaining, 1);
    }
    negative_cols = (unsigned long long) wave_max_i64((long long) negative_cols);
    if (lane == 0 && negative_cols)
        st->error_b += (long long) negative_cols;
}

template<int RPL>
__global__ __launch_bounds__(128) void k_insert2(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot,
                                                 const uint8_t* __restrict__ inten, long long n, int* remaining, long long n_total, long long fbase)
{
    insert2_body<RPL>(g, cfg, P, states, first_stream, slot, inten, n, remaining, n_total, fbase, (int) blockIdx.x);
}

// ---- pieces shared by k_insert_par and k_insert_par_fin ---------------------------------------------------------------------------
// take back what firings behind the first offending one have written: every cell of the columns (rel_from .. rel_to past prev_rear0) returns
// to the cleared state (clearColumns' three planes: all the serial kernel looks at). Whole columns: with the fused segmentation cells without
// a return carry the ring-pass tag as well.
template<int RPL>
__device__ __forceinline__ void par_take_back(const SP& p, const int R, const int RC, const int lc0, const int rel_from, const int rel_to, const int wave,
                                              const int nwaves, const int lane)
{
    for (int rel = rel_from + wave; rel <= rel_to; rel += nwaves)
    {
        const int lc = (int) ((unsigned) (lc0 + rel) % (unsigned) RC);
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            if (row < R)
            {
                const size_t ci = (size_t) lc * R + row;
                p.dist[ci] = __builtin_nanf("");
                p.incl[ci] = __builtin_nanf("");
                p.gtag[ci] = CELL_CLEARED;
            }
        }
    }
}

// the stream's state behind a run of `done` firings (one thread); returns whether the batch is closed as FUSED
__device__ __forceinline__ int par_close_stream(StreamState* st, const int slot, int* left_over, const bool fuse, const bool whole, const int done,
                                                const long long n, const long long prev_rear0, const long long first_unf0, const long long ring_end0,
                                                const long long seq0, const long long rel_last)
{
    (void) n;
    int fused = 0;
    if (done > 0)
    {
        const long long G = prev_rear0 + rel_last;
        st->prev_rearmost = G;
        st->prev_foremost = G;
        st->first_unfinished = G;
        if (G > ring_end0)
            st->ring_end = G;
        st->cursor = done;
        st->firings_consumed = (unsigned long long) (seq0 + done);
        st->pre_seg_begin = first_unf0;
    }
    // left_over (the engine's "skip_idle_fallbacks"): the host launches the other insertion kernels of this batch only if some stream
    // needs them. A stream whose whole batch went through here closes its batch descriptor itself, exactly as k_insert2 would with
    // nothing left to do (its columns [first_unf0, G) were emitted, cursor = n).
    if (left_over)
    {
        if (whole)
        {
            const long long G = prev_rear0 + rel_last;
            fused = fuse && ld_agent(&st->error) == 0 ? 1 : 0;
            st->batch[slot].seg_begin = first_unf0;
            st->batch[slot].seg_end = G;
            st->batch[slot].acp_next = first_unf0;
            st->batch[slot].pub_begin = -1;
            st->batch[slot].pub_end = -1;
            st->batch[slot].fused = fused;
#if !defined(CC_PROFILE_SECTIONS) && !defined(CC_A2_STATS)
            st->dbg[4] += (unsigned long long) fused; // batches closed as fused (cc_engine_debug_counters; tests)
#endif
            if (fused)
                st->batch[slot].mode = st->assoc_mode; // (what k_table does first for the streams it sees)
            // (pre_seg_begin stays: if another stream makes the host launch the other insertion kernels after all, k_insert2 finds
            // nothing left for this stream and leaves the descriptor alone; k_begin_batch clears it for the next batch)
            if (!fused)
                atomicAdd(left_over + 1, 1); // streams whose batch still needs k_table / k_seg_pre
        }
        else
        {
            atomicAdd(left_over, 1);
            atomicAdd(left_over + 1, 1);
        }
    }
    return fused;
}

// k_table's phase 2 from the partials the wavefronts of the fused insertion left in Planes::tab_acc (one wavefront, lanes = rows): the tiles'
// entries become the table in front of each tile (Planes::tabc) and the stream's table moves on — or, when the batch is not closed as fused,
// the partials are only wiped (k_table will read the columns from the ring). `touched` tiles may hold partials, `ntiles` are the batch's.
template<int RPL>
__device__ __forceinline__ void table_from_partials(const SP& p, const int R, const bool fused, const int touched, const int ntiles, const int lane)
{
    float carry[RPL];
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        const int row = k * 64 + lane;
        carry[k] = (fused && row < R) ? p.curtab[row] : 0.f;
    }
    constexpr int U = 8;
    for (int t0 = 0; t0 < touched; t0 += U)
    {
        unsigned long long v[U][RPL];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                v[u][k] = (row < R && t0 + u < touched) ? p.tab_acc[(size_t) (t0 + u) * R + row] : 0ull;
            }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (row < R && t0 + u < touched)
                {
                    if (v[u][k])
                        p.tab_acc[(size_t) (t0 + u) * R + row] = 0ull;
                    if (fused && t0 + u < ntiles)
                    {
                        p.tabc[(size_t) (t0 + u) * R + row] = carry[k];
                        if (v[u][k] >> 32)
                            carry[k] = __uint_as_float((unsigned) v[u][k]);
                    }
                }
            }
    }
    if (fused)
    {
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            if (row < R)
                p.curtab[row] = carry[k];
        }
    }
}

// =====================================================================================================
// k_insert_par — insertFiringIntoRangeImage (cc.cpp:105-292) for the head of a batch, all firings at once, straight from the
// caller's buffers (the per-point preparation is done inline: what this kernel takes never touches the staging planes).
//
// The serial recurrence of the insertion is "global column of this firing relative to the previous rearmost laser". For the firing
// shape the reference's own harness produces (kd.cpp:123-159: every return of a firing in one column) and a sensor that advances by
// at least one column per firing, that recurrence is a prefix sum: with c_f the column-in-rotation of firing f, the global column is
// G_f = G_(f-1) + d_f, d_f = c_f - c_(f-1) (+ num_columns across the rotation wrap, cc.cpp:165-175), and under d_f > 0 the rearmost =
// foremost = G_f, nothing is "too far behind", firing f finishes exactly the columns [G_(f-1), G_f) (cc.cpp:289-291), and no target
// cell can be occupied: nothing was ever written ahead of the foremost laser, and the previous tenant of the ring slot, column
// G_f - ring_cols, has been cleared when it lies below StreamState::clear_done. One block per stream:
//   0  one lane per firing: the column c_f of its first valid return (one atan2f per firing)
//   B  block scan of d_f -> G_f for the whole batch; the first firing that breaks a condition (empty firing, d_f <= 0 or backwards,
//      emission limit, ring slot not provably clear) ends the run
//   D  wave per firing, no barriers: rigid transform, range, azimuth, inclination of its returns, the nine planes of its cells, the
//      finishing firing of the columns it completes. The one condition only this phase can see — a return in another column than
//      the firing's first — is rare; the run then ends at that firing and whatever later firings have already written is taken back
//      (their cells return to the cleared state, which is all the serial kernel looks at).
// The rest of the batch (from the first firing that does not fit: a multi-column sensor, a stream that is not in steady state yet,
// two firings in one column ...) goes to k_prep + k_insert2 through StreamState::cursor, with exactly the state the serial kernel
// would have at that firing. grid = streams, block = 64 * IP_WAVES.
// =====================================================================================================
// 8 wavefronts per block: alone the kernel is faster with 16 (0.70 vs 0.8 ms), but in the pipeline it shares every CU with the
// segmentation / scan kernels, and the step is 4 % shorter when it holds half the registers and wave slots
#ifndef CC_IP_WAVES
#define CC_IP_WAVES 8
#endif
constexpr int IP_WAVES = CC_IP_WAVES;

// (W wavefronts per block: IP_WAVES next to the other chains' kernels; twice as many when a launch has few streams and the GPU is otherwise empty)
template<int RPL, int W = IP_WAVES>
__global__ __launch_bounds__(64 * W) void k_insert_par(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream,
                                                            const float* __restrict__ xyz, const uint8_t* __restrict__ inten,
                                                            const double* __restrict__ poses, long long n, long long n_total, long long fbase,
                                                            int slot, int* __restrict__ left_over, const double* __restrict__ ego)
{
    const int sl = blockIdx.x;
    const int s = first_stream + sl;
    const int lane = lane_id();
    const int wave = uniform_i32((int) (threadIdx.x >> 6)); // (readfirstlane: the firing index and everything addressed with it stay scalar)
    const int tid = threadIdx.x;
    StreamState* st = &states[s];
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, NC = g.num_columns, RC = g.ring_cols;
    // FUSED SEGMENTATION (round 4): a firing of the run fills its column alone and the next firing finishes it, so the wavefront that has the
    // column's cells in registers also does the per-cell part of its ground segmentation (seg_pre_cells: what k_seg_pre would read back from
    // the ring) with the NEXT firing's pose (the job's pose, cc.cpp:291) and leaves each tile's last valid inclination step (k_table's phase 1)
    // in Planes::tab_acc. When the whole batch is taken that way the batch descriptor says so (BatchDesc::fused) and neither k_table nor
    // k_seg_pre has anything to do for the stream; otherwise they redo the batch's columns from the ring as before (everything written here
    // is what they would write, or is overwritten by them). Needs the gate (left_over) and the per-firing records of k_ego.
    const bool fuse = left_over != nullptr && ego != nullptr && st->has_robot_tf != 0;
    __shared__ short s_c[IP_MAXF]; // column-in-rotation of every firing (its first valid return), -1 = empty firing (or a column index above 32767: the
                                   // run ends there and the serial kernel takes over — 9 KB less LDS for a block that has to find room next to the other chains)
    __shared__ unsigned short s_off[IP_MAXF]; // G_f - prev_rearmost at entry (a firing more than 65535 columns ahead of it ends the run)
    __shared__ int s_wsum[W];
    __shared__ int s_upto, s_bad, s_carry;

    const long long prev_rear0 = st->prev_rearmost, prev_fore0 = st->prev_foremost, first_unf0 = st->first_unfinished;
    const long long ring_end0 = st->ring_end;
    // ring_start belongs to the association chain, which may be advancing it right now (previous batch): every wavefront has to work
    // with the same value, or the columns between two wavefronts' views would be skipped by the clearing below
    __shared__ long long s_ring_start;
    if (tid == 0)
        s_ring_start = __hip_atomic_load(&st->ring_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const long long ring_start = s_ring_start;
    // deferred clearColumns (cc.cpp:1094-1145) exactly as k_insert2 would do it first, spread over the wavefronts
    // gridDim.y > 1: the firings of the stream are dealt to several blocks (few streams on a big GPU). Every block repeats phases 0 and B (cheap), block
    // 0 clears, nobody writes the stream state: k_insert_par_fin does that once all blocks are through.
    const int by = (int) blockIdx.y, nby = (int) gridDim.y;
    long long clear_done = st->clear_done;
    const long long clear_done_entry = clear_done;
    if (clear_done >= 0 && by == 0)
    {
        const long long clear_to = ring_start < st->clear_allowed ? ring_start : st->clear_allowed;
        for (long long c = clear_done + wave; c < clear_to; c += W)
        {
            const int clc = (int) (c % RC);
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (row < R)
                {
                    const size_t ci = (size_t) clc * R + row;
                    p.dist[ci] = __builtin_nanf("");
                    p.incl[ci] = __builtin_nanf("");
                    p.gtag[ci] = CELL_CLEARED;
                }
            }
        }
        if (clear_to > clear_done)
            clear_done = clear_to;
    }
    const bool steady = st->cursor == 0 && ring_start != -1 && first_unf0 > 0 && first_unf0 == prev_rear0 && prev_fore0 == prev_rear0 &&
                        st->reset_required == 0 && st->pre_seg_begin == 0 && clear_done >= 0;
    const int nn = (int) (n < IP_MAXF ? n : IP_MAXF);
    if (tid == 0)
    {
        s_upto = nn;
        s_carry = 0;
    }
    __syncthreads(); // the cleared cells are ordered before everything this block writes from here on
    if (!steady)
    {
        if (tid == 0 && by == 0)
        {
            st->clear_done = clear_done;
            if (left_over)
            {
                atomicAdd(left_over, 1); // the other insertion kernels have to take this stream's batch
                atomicAdd(left_over + 1, 1); // ... and k_table / k_seg_pre its segmentation
            }
            st->par_upto = -1;
        }
        return;
    }
    // (split over blocks: every block has to end the run at the same firing, so the "previous tenant of the ring slot is cleared" test uses what
    // was cleared BEFORE this launch — block 0 clears columns >= that, accepted firings only touch slots whose previous tenant lies below it)
    const long long clear_known = nby > 1 ? clear_done_entry : clear_done;
    const int half = NC / 2;
    const bool clockwise = cfg.sensor_is_clockwise != 0;
    const size_t fglob = (size_t) sl * (size_t) n_total + (size_t) fbase; // first firing of this batch in the caller's buffers
    const long long seq0 = (long long) st->firings_consumed;
    const long long rot0 = prev_rear0 / NC;
    const int cir0 = (int) (prev_rear0 - rot0 * NC);
    const int lc0 = (int) (prev_rear0 % RC);
    const long long pass0 = prev_rear0 / RC; // pass over the ring of the previous rearmost laser (cell_tag)

    // ---- 0: the column of every firing from its first valid return (prep_point's column arithmetic, nothing else of it)
    for (int f = tid; f < nn; f += 64 * W)
    {
        const size_t base = (fglob + (size_t) f) * R * 3;
        int c = -1;
        for (int row = 0; row < R; row++)
        {
            const float fx = xyz[base + (size_t) row * 3];
            if (fx == fx)
            {
                const float fy = xyz[base + (size_t) row * 3 + 1];
                const float az = ccm::atan2f_exact(fy, fx);
                const float inc_az = clockwise ? -az + CC_PI_F : az + CC_PI_F;
                c = f2i_x86(inc_az / g.az_width);
                break;
            }
        }
        s_c[f] = (short) ((c >= 0 && c < NC && c < 32768) ? c : -1);
    }
    __syncthreads();
    // ---- B: column advance of every firing, its prefix sum over the batch, first firing that ends the run
    for (int base = 0; base < nn; base += 64 * W)
    {
        const int f = base + tid;
        const int c = f < nn ? s_c[f] : -1;
        const int cp = f == 0 ? cir0 : (f < nn ? s_c[f - 1] : -1);
        const int diff = c - cp;
        // strictly forward, also across the wrap (cc.cpp:165-175)
        const bool ok = f < nn && c >= 0 && cp >= 0 && ((diff > 0 && diff <= half) || diff < -half);
        const int delta = ok ? (diff < -half ? diff + NC : diff) : 0;
        int v = delta;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1)
        {
            const int o = __shfl_up(v, d, 64);
            if (lane >= d)
                v += o;
        }
        if (lane == 63)
            s_wsum[wave] = v;
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wave; w++)
            before += s_wsum[w];
        const int incl = before + v;
        if (f < nn)
        {
            s_off[f] = (unsigned short) incl;
            const long long G = prev_rear0 + incl;
            const long long rear_before = G - delta;
            // a firing is only taken while the batch has emitted fewer than limit_columns columns before it (k_insert2's loop head), and
            // while the previous tenant of its ring slot is known to be cleared
            if (!ok || incl > 65535 || rear_before - first_unf0 >= g.limit_columns || G - RC >= clear_known)
                atomicMin(&s_upto, f);
        }
        __syncthreads();
        if (tid == 64 * W - 1)
            s_carry = incl;
        __syncthreads();
    }
    const int upto = s_upto;
    if (tid == 0)
        s_bad = upto;
    __syncthreads();
    // ---- D: the cells and the columns each firing finishes; wavefronts run independently. A wavefront's firings are latency chains
    // (load the returns -> ~300 instructions of arithmetic -> store the cells) and there are only two wavefronts per SIMD to hide
    // them, so the inputs of the wavefront's NEXT firing (returns, intensities, pose: one lane per matrix element) are loaded before
    // the current one is worked on.
    float nx_x[RPL], nx_y[RPL], nx_z[RPL];
    uint8_t nx_i[RPL];
    auto load_firing = [&](const int f)
    {
        const size_t fi = fglob + (size_t) f;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            nx_x[k] = nx_y[k] = nx_z[k] = __builtin_nanf("");
            nx_i[k] = 0;
            if (row < R && f < upto)
            {
                const size_t src = (fi * R + row) * 3;
                nx_x[k] = xyz[src];
                nx_y[k] = xyz[src + 1];
                nx_z[k] = xyz[src + 2];
                nx_i[k] = inten[fi * R + row];
            }
        }
    };
    const int fstep = W * nby;
#ifdef CC_IP_PRIO
    __builtin_amdgcn_s_setprio(CC_IP_PRIO); // (experiment switch: issue priority of the insertion's wavefronts next to the other chains' kernels)
#endif
    // ---- fused segmentation: per-wavefront partial of k_table's phase 1 (a wavefront's columns increase: the last valid step it has seen in the
    // tile it is in; flushed into Planes::tab_acc with an atomic max on (column, step) when it moves on to another tile)
    float tl_val[RPL];
    int tl_col[RPL], tl_tile = -1;
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        tl_val[k] = 0.f;
        tl_col[k] = 0;
    }
    auto tl_flush = [&]()
    {
        if (tl_tile >= 0)
        {
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (row < R && tl_col[k] > 0)
                    atomicMax(&p.tab_acc[(size_t) tl_tile * R + row], ((unsigned long long) (unsigned) tl_col[k] << 32) | (unsigned long long) __float_as_uint(tl_val[k]));
                tl_col[k] = 0;
            }
        }
    };
    // staging of one segmented column (column `rel` columns past prev_rear0): the per-cell results, the ring-pass tag and a record for EVERY cell
    // (cells without a return: the NaN record k_seg_scan completes with the supplemented inclination), the column's entries, its table partial
    const float rcp_rc = 1.0f / (float) RC, rcp_nc = 1.0f / (float) NC;
    // x / d for x < 2^17 (columns past prev_rear0 plus a ring / rotation offset): float estimate, corrected — a hardware-free 32-bit division
    // costs ~25 instructions, and two of them per firing were 6 % of this kernel
    auto div_small = [](const int x, const int d, const float rcp, int& rem) -> int
    {
        int q = (int) ((float) x * rcp);
        int r = x - q * d;
        if (r < 0)
        {
            q--;
            r += d;
        }
        else if (r >= d)
        {
            q++;
            r -= d;
        }
        rem = r;
        return q;
    };
    CazBase cbw = caz_base_of_rotation(rot0);
    int cbw_rot = 0; // rotations past rot0 the cached base belongs to
    auto stage_column = [&](const int rel, const float (&x2)[RPL], const float (&uz)[RPL], const float (&w)[RPL], const int (&flags)[RPL],
                            const float (&incaz)[RPL], const bool write_empty_cells)
    {
        int lc;
        const int lcq = div_small(lc0 + rel, RC, rcp_rc, lc);
        const uint16_t tag = cell_tag(pass0 + (long long) lcq);
        const long long G = prev_rear0 + rel;
        int cirg;
        const int rq = div_small(cir0 + rel, NC, rcp_nc, cirg);
        if (rq != cbw_rot) // (wave-uniform; once per rotation)
        {
            cbw = caz_base_of_rotation(rot0 + (long long) rq);
            cbw_rot = rq;
        }
        const int tile = rel >> 6;
        if (tile != tl_tile)
        {
            tl_flush();
            tl_tile = tile;
        }
        int kpos = 0x7fffffff, kneg = 0x7fffffff;
        bool any_empty = false;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            if (row >= R)
                continue;
            const unsigned ci = (unsigned) lc * (unsigned) R + (unsigned) row;
            at32(p.sg_x2, ci) = x2[k];
            at32(p.sg_uz, ci) = uz[k];
            at32(p.sg_w, ci) = w[k];
            at32(p.sg_flags, ci) = (uint8_t) flags[k];
            if ((flags[k] & SG_NAN) && write_empty_cells)
            {
                at32(p.gtag, ci) = tag;
                at32(p.sc_rec, ci) = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            }
            if (flags[k] & SG_NAN)
                any_empty = true;
            else
                caz_key(incaz[k], kpos, kneg);
            if (!(flags[k] & (SG_NAN | SG_PENDING)))
            {
                tl_val[k] = w[k];
                tl_col[k] = rel + 1;
            }
        }
        const double min_az = column_min_caz(cbw, kpos, kneg, any_empty, G, g.az_width);
        if (lane == 0)
        {
            p.colg[lc] = G;
            p.colminaz[lc] = min_az;
        }
    };
    // the columns (from, to) past prev_rear0 that no firing fills (the sensor skipped them): segmented as columns without returns
    auto stage_gap = [&](const int from, const int to)
    {
        for (int rel = from; rel < to; rel++)
        {
            float x2[RPL], uz[RPL], w[RPL], az[RPL];
            int flags[RPL];
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                x2[k] = uz[k] = az[k] = 0.f;
                w[k] = __builtin_nanf("");
                flags[k] = SG_NAN;
            }
            stage_column(rel, x2, uz, w, flags, az, true);
        }
    };
    if (fuse && by == 0 && wave == 0 && upto > 0)
    {
        // the column the previous batch left open (prev_rear0 = first_unf0, cells in the ring) is finished by this batch's first firing
        const uint16_t tag = cell_tag(pass0);
        float cx[RPL], cy[RPL], cz[RPL], dist[RPL], incl[RPL], az[RPL];
        uint8_t it[RPL];
        bool overrun = false;
        int overrun_row = -1;
        long long overrun_gcol = -1;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            cx[k] = cy[k] = cz[k] = az[k] = 0.f;
            dist[k] = incl[k] = __builtin_nanf("");
            it[k] = 0;
            if (row < R)
            {
                const size_t ci = (size_t) lc0 * R + row;
                const uint16_t tg = p.gtag[ci];
                dist[k] = p.dist[ci];
                if (tg == tag)
                {
                    const float4 r4 = p.sc_rec[ci];
                    cx[k] = r4.x, cy[k] = r4.y, cz[k] = r4.z, incl[k] = r4.w;
                    az[k] = p.incaz[ci];
                    it[k] = p.inten[ci];
                }
                else if (tg != CELL_CLEARED)
                {
                    overrun = true; // cc.cpp:320-345 (as in k_seg_pre)
                    overrun_row = row;
                    overrun_gcol = prev_rear0 - (long long) ((((unsigned) tag - (unsigned) tg) & 0x7fffu)) * RC;
                }
                else
                {
                    p.gtag[ci] = tag; // (as the segmentation tags a cell without a return, cc.cpp:348-351 — with the record such a cell has)
                    p.sc_rec[ci] = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                }
            }
        }
        if (__any(overrun))
        {
            const int worst = -wave_min_i32(-overrun_row);
            if (overrun_row == worst)
            {
                atomicMin((unsigned long long*) &st->overrun_col, (unsigned long long) prev_rear0);
                raise_error(st, CC_ERR_RING_OVERRUN, overrun_gcol, prev_rear0);
            }
        }
        else
        {
            const double* T0 = poses + fglob * 12;
            const double* E0 = ego + ((size_t) sl * (size_t) n) * EGO_STRIDE;
            float x2[RPL], uz[RPL], w[RPL];
            int flags[RPL];
            seg_pre_cells<RPL>(cfg, R, lane, cx, cy, cz, dist, incl, it, (float) T0[3], (float) T0[7], (float) T0[11], E0, x2, uz, w, flags);
            stage_column(0, x2, uz, w, flags, az, false);
        }
        stage_gap(1, (int) s_off[0]);
    }
    if (wave + W * by < upto)
        load_firing(wave + W * by);
    for (int f = wave + W * by; f < upto; f += fstep)
    {
        float cx[RPL], cy[RPL], cz[RPL];
        uint8_t cint[RPL];
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            cx[k] = nx_x[k];
            cy[k] = nx_y[k];
            cz[k] = nx_z[k];
            cint[k] = nx_i[k];
        }
        // the firing's pose by SCALAR loads (f is wave-uniform): as a vector load with one lane per matrix element, prefetched with the returns, the
        // matrix cost 30 v_readlane per firing on a GPU whose vector ALUs are what the step waits for; the scalar loads' latency is other wavefronts' time
        const double* Tp = poses + (fglob + (size_t) f) * 12;
        double T[12]; // (wave-uniform: the matrix travels in SGPRs)
#pragma unroll
        for (int i = 0; i < 12; i++)
            T[i] = Tp[i];
        // translation of the NEXT firing's pose = sgps_sensor_position of this column's job
        const bool has_next = fuse && f + 1 < upto;
        const float spx = has_next ? (float) Tp[12 + 3] : 0.f, spy = has_next ? (float) Tp[12 + 7] : 0.f, spz = has_next ? (float) Tp[12 + 11] : 0.f;
        load_firing(f + fstep);
        if (f > lds_ld(&s_bad)) // some earlier firing left the shape: nothing behind it is wanted (wave-uniform)
            break;
        const size_t fi = fglob + (size_t) f;
        const int c0 = s_c[f];
        PreppedPoint q[RPL];
        bool differs = false;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            q[k].cir = PP_SKIP;
            if (row < R)
                q[k] = prep_point(cx[k], cy[k], cz[k], T, clockwise, g.az_width);
            differs |= q[k].cir != PP_SKIP && q[k].cir != c0;
        }
        if (__any(differs))
        {
            if (lane == 0)
                atomicMin(&s_bad, f);
            break; // this wavefront's later firings lie behind it
        }
        const long long rel = s_off[f];                    // G_f - prev_rear0
        const long long rel_prev = f > 0 ? s_off[f - 1] : 0; // G_(f-1) - prev_rear0
        const long long G = prev_rear0 + rel;
        int lc;
        const int lcq = div_small(lc0 + (int) rel, RC, rcp_rc, lc); // (quotient = passes over the ring since lc0)
        const uint16_t tag = cell_tag(pass0 + (long long) lcq);
        const bool seg_here = fuse && f + 1 < upto; // (the run's last firing leaves its column open: nobody has finished it yet)
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            const unsigned ci = (unsigned) lc * (unsigned) R + (unsigned) row;
            const bool has = q[k].cir != PP_SKIP;
            // a cell without a return of a column segmented here is tagged like the segmentation tags it (cc.cpp:348-351) and gets the record of a
            // cell without a return (k_seg_scan completes it with the supplemented inclination): one store each for all the column's cells
            if (has | (seg_here & (row < R)))
            {
                const float nn = __builtin_nanf("");
                at32(p.sc_rec, ci) = make_float4(has ? q[k].x : nn, has ? q[k].y : nn, has ? q[k].z : nn, has ? q[k].incl : nn);
                at32(p.gtag, ci) = tag;
            }
            if (has)
            {
                at32(p.inten, ci) = cint[k];
                at32(p.src, ci) = (uint32_t) (seq0 + f);
                at32(p.dist, ci) = q[k].dist;
                at32(p.incl, ci) = q[k].incl;
                at32(p.incaz, ci) = q[k].incaz; // (c0 < num_columns and nothing moves on: the return's rotation is its column's)
            }
        }
        // columns [G_(f-1), G_f) are finished by this firing and carry its pose (cc.cpp:289-291)
        const int cnt = (int) (rel - rel_prev);
        for (int jj = lane; jj < cnt; jj += 64)
        {
            int tlc;
            (void) div_small(lc0 + (int) rel_prev + jj, RC, rcp_rc, tlc);
            p.trig[tlc] = f;
        }
        if (seg_here)
        {
            // the per-cell part of this column's ground segmentation; its job carries the NEXT firing's pose
            float sx[RPL], sy[RPL], sz[RPL], sd[RPL], si_[RPL], saz[RPL], x2[RPL], uz[RPL], w[RPL];
            int flags[RPL];
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const bool has = q[k].cir != PP_SKIP;
                sx[k] = q[k].x, sy[k] = q[k].y, sz[k] = q[k].z, saz[k] = q[k].incaz;
                sd[k] = has ? q[k].dist : __builtin_nanf("");
                si_[k] = has ? q[k].incl : __builtin_nanf("");
            }
            // (f is wave-uniform, but only readfirstlane tells the compiler: the record then arrives by SCALAR loads — as vector loads its first
            // word cost an s_waitcnt vmcnt(0) per firing, i.e. a wait for every store of the previous firing)
            const double* E = ego + ((size_t) sl * (size_t) n + (size_t) (uniform_i32(f) + 1)) * EGO_STRIDE;
            seg_pre_cells<RPL>(cfg, R, lane, sx, sy, sz, sd, si_, cint, spx, spy, spz, E, x2, uz, w, flags);
            stage_column((int) rel, x2, uz, w, flags, saz, false);
            stage_gap((int) rel + 1, (int) s_off[f + 1]);
        }
    }
    if (fuse)
        tl_flush();
    __syncthreads();
    if (nby > 1)
    {
        // several blocks per stream: leave the offsets and the two ends of the run for k_insert_par_fin
        if (s_bad < upto && tid == 0)
            atomicMin(&st->par_bad, s_bad);
        if (by == 0)
        {
            for (int f = tid; f < upto; f += 64 * W)
                p.par_off[f] = s_off[f];
            if (tid == 0)
            {
                st->par_upto = upto;
                st->par_clear_done = clear_done;
                if (upto == 0 && left_over)
                {
                    atomicAdd(left_over, 1); // (nothing taken: k_insert_par_fin has nothing to do either)
                    atomicAdd(left_over + 1, 1);
                }
            }
        }
        return;
    }
    const int done = s_bad < upto ? s_bad : upto;
    if (done < upto)
        par_take_back<RPL>(p, R, RC, lc0, (int) s_off[done], (int) s_off[upto - 1], wave, W, lane);
    const bool whole = done == (int) n && done > 0;
    __shared__ int s_fused;
    if (tid == 0)
    {
        st->clear_done = clear_done;
#ifndef CC_A2_STATS
        st->dbg[6] += (unsigned long long) done; // firings taken by this kernel / batches it saw (cc_engine_debug_counters)
        st->dbg[7] += 1;
#endif
        s_fused = par_close_stream(st, slot, left_over, fuse, whole, done, n, prev_rear0, first_unf0, ring_end0, seq0,
                                   done > 0 ? (long long) s_off[done - 1] : 0);
    }
    __syncthreads(); // (also: every wavefront's table partials have reached Planes::tab_acc)
    if (fuse && wave == 0 && upto > 0)
        table_from_partials<RPL>(p, R, s_fused != 0, ((int) s_off[upto - 1] >> 6) + 1, s_fused ? (int) ((s_off[done - 1] + 63) >> 6) : 0, lane);
}

// k_insert_par_fin — what one block of k_insert_par does behind its phase D, for launches that dealt a stream's firings to several blocks: take back
// what lies behind the first offending firing, then the stream state, the batch descriptor and (fused segmentation) the table. grid = streams, block = 256.
template<int RPL>
__global__ __launch_bounds__(256) void k_insert_par_fin(Geometry g, Planes P, StreamState* states, int first_stream, const float* __restrict__ xyz,
                                                        long long n, long long n_total, long long fbase, int slot, int* __restrict__ left_over, int fuse_on)
{
    (void) xyz;
    (void) n_total;
    (void) fbase;
    const int sl = blockIdx.x;
    const int s = first_stream + sl;
    const int lane = lane_id(), wave = uniform_i32((int) (threadIdx.x >> 6)), tid = threadIdx.x;
    StreamState* st = &states[s];
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols;
    const int upto = st->par_upto;
    if (upto <= 0)
    {
        if (upto == 0 && tid == 0 && st->par_clear_done >= 0)
            st->clear_done = st->par_clear_done;
        return; // (not steady, or nothing taken: block 0 of k_insert_par has counted the stream as left over)
    }
    const bool fuse = fuse_on != 0 && left_over != nullptr && st->has_robot_tf != 0;
    const int bad = st->par_bad;
    const int done = bad < upto ? bad : upto;
    const long long prev_rear0 = st->prev_rearmost, first_unf0 = st->first_unfinished, ring_end0 = st->ring_end;
    const int lc0 = (int) (prev_rear0 % RC);
    const long long seq0 = (long long) st->firings_consumed;
    if (done < upto)
        par_take_back<RPL>(p, R, RC, lc0, p.par_off[done], p.par_off[upto - 1], wave, 4, lane);
    const bool whole = done == (int) n && done > 0;
    __shared__ int s_fused;
    __syncthreads(); // (everybody has read the state thread 0 is about to replace)
    if (tid == 0)
    {
        st->clear_done = st->par_clear_done;
#ifndef CC_A2_STATS
        st->dbg[6] += (unsigned long long) done;
        st->dbg[7] += 1;
#endif
        s_fused = par_close_stream(st, slot, left_over, fuse, whole, done, n, prev_rear0, first_unf0, ring_end0, seq0, done > 0 ? (long long) p.par_off[done - 1] : 0);
    }
    __syncthreads();
    if (fuse && wave == 0)
        table_from_partials<RPL>(p, R, s_fused != 0, (p.par_off[upto - 1] >> 6) + 1, s_fused ? ((p.par_off[done - 1] + 63) >> 6) : 0, lane);
}

// =====================================================================================================
// k_insert_multi — the block-parallel insertion for MULTI-COLUMN firings (sensors whose lasers carry individual azimuth offsets: a
// VLS-128 firing spans ~60 columns; cc.cpp:105-292), and for whatever single-column head k_insert_par left over.
//
// What makes the insertion serial is (1) the column of every return relative to the previous rearmost laser (cc.cpp:152-175) and (2) the
// per-row collision rule (cc.cpp:188-206). With r_f the column-in-rotation of the REARMOST laser of firing f, (1) is again a prefix sum
// while the rearmost laser advances by >= 1 column per firing: rear column G_f = G_(f-1) + unwrap(r_f - r_(f-1)), and a return whose
// column-in-rotation lies o columns ahead of r_f lands in column G_f + o. (2) never fires while every ROW's target columns increase
// strictly from firing to firing (a cell of row i can only have been written by an earlier return of row i: rows never share cells) and
// the previous tenant of the ring slot has been cleared. Both conditions are CHECKED, per firing and per row, before anything is written;
// the first firing that violates one (empty firing, rearmost laser not advancing, a row revisiting or falling behind one of its earlier
// columns, span of half a rotation, ring slot not provably clear, emission limit) ends the run and the serial kernel continues there,
// with exactly the state it would have at that firing. No roll-back is needed: nothing of a firing is written before it is accepted.
//
// One block per stream, IM_WAVES wavefronts, chunks of IM_WAVES firings: every wavefront prepares one firing (rigid transform, range,
// bit-exact atan2f / asinf: prep_point) and keeps its points in registers, the chunk's rear columns and per-row target columns meet in LDS
// (two barriers per chunk), then every accepted firing writes its cells. grid = streams, block = 64 * IM_WAVES.
// =====================================================================================================
#ifndef CC_IM_WAVES
#define CC_IM_WAVES 8
#endif
constexpr int IM_WAVES = CC_IM_WAVES;

template<int RPL>
__global__ __launch_bounds__(64 * IM_WAVES) void k_insert_multi(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream,
                                                              const float* __restrict__ xyz, const uint8_t* __restrict__ inten,
                                                              const double* __restrict__ poses, long long n, long long n_total, long long fbase,
                                                              int slot, int* __restrict__ left_over)
{
    // left_over (engine option skip_idle_fallbacks, launches in which this is the first insertion kernel): a stream whose whole batch is taken here
    // gets its batch descriptor here (as in k_insert_par); every other stream is counted, and the host launches k_prep + k_insert2 only if there is one
    const int sl = blockIdx.x;
    const int s = first_stream + sl;
    const int lane = lane_id();
    const int wave = uniform_i32((int) (threadIdx.x >> 6));
    const int tid = threadIdx.x;
    StreamState* st = &states[s];
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, NC = g.num_columns, RC = g.ring_cols;
    constexpr int NR = 64 * RPL;
    __shared__ int s_rear_cir[IM_WAVES]; // column-in-rotation of the firing's rearmost laser, -1 = empty firing
    __shared__ int s_span[IM_WAVES];     // foremost - rearmost column of the firing
    __shared__ int s_col[IM_WAVES][NR];  // per row: columns ahead of the firing's rearmost laser, -1 = no return
    __shared__ int s_rowmax[NR];         // per row: last column written (relative to prev_rearmost at entry), INT_MIN = none in reach
    __shared__ int s_stop;               // first firing of the chunk whose rows clash with earlier returns
    __shared__ long long s_ring_start;

    const long long cursor0 = st->cursor;
    if (cursor0 >= n)
    {
        if (left_over && tid == 0)
            atomicAdd(left_over, 1); // (an empty call, or a batch another kernel closed: the serial kernel writes the descriptor)
        return;
    }
    const long long prev_rear0 = st->prev_rearmost, prev_fore0 = st->prev_foremost, first_unf0 = st->first_unfinished;
    const long long ring_end0 = st->ring_end;
    if (tid == 0)
        s_ring_start = __hip_atomic_load(&st->ring_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const long long ring_start = s_ring_start;
    // deferred clearColumns (cc.cpp:1094-1145), spread over the wavefronts (as in k_insert_par; nothing left to do when that kernel ran)
    long long clear_done = st->clear_done;
    if (clear_done >= 0)
    {
        const long long clear_to = ring_start < st->clear_allowed ? ring_start : st->clear_allowed;
        for (long long c = clear_done + wave; c < clear_to; c += IM_WAVES)
        {
            const int clc = (int) (c % RC);
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (row < R)
                {
                    const size_t ci = (size_t) clc * R + row;
                    p.dist[ci] = __builtin_nanf("");
                    p.incl[ci] = __builtin_nanf("");
                    p.gtag[ci] = CELL_CLEARED;
                }
            }
        }
        if (clear_to > clear_done)
            clear_done = clear_to;
    }
    const bool steady = ring_start != -1 && first_unf0 > 0 && first_unf0 == prev_rear0 && prev_fore0 >= prev_rear0 && st->reset_required == 0 &&
                        clear_done >= 0 && prev_fore0 - prev_rear0 < NC / 2;
    __syncthreads(); // the cleared cells are ordered before everything this block writes from here on
    if (!steady)
    {
        if (tid == 0)
        {
            st->clear_done = clear_done;
            if (left_over)
                atomicAdd(left_over, 1);
        }
        return;
    }
    // what the rows have written ahead of the rearmost laser so far: the last occupied column of every row in [prev_rear0, prev_fore0]
    for (int r = tid; r < NR; r += 64 * IM_WAVES)
        s_rowmax[r] = (int) 0x80000000;
    __syncthreads();
    {
        const int ahead = (int) (prev_fore0 - prev_rear0);
        const int lc_base = (int) (prev_rear0 % RC);
        for (int c = wave; c <= ahead; c += IM_WAVES)
        {
            int lc = lc_base + c;
            lc = lc >= RC ? lc - RC : lc;
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (row < R)
                {
                    const float d = p.dist[(size_t) lc * R + row];
                    if (d == d)
                        atomicMax(&s_rowmax[row], c);
                }
            }
        }
    }
    __syncthreads();

    const int half = NC / 2;
    const bool clockwise = cfg.sensor_is_clockwise != 0;
    const size_t fglob = (size_t) sl * (size_t) n_total + (size_t) fbase;
    const long long seq0 = (long long) st->firings_consumed;
    const long long rot0 = prev_rear0 / NC;
    const int cir0 = (int) (prev_rear0 - rot0 * NC);
    const int lc0 = (int) (prev_rear0 % RC);
    const long long pass0 = prev_rear0 / RC; // pass over the ring of the previous rearmost laser (cell_tag)
    // carried from chunk to chunk (every thread keeps the same values)
    int carry_rel = 0;    // rear column of the last accepted firing, relative to prev_rear0
    int carry_cir = cir0; // its column-in-rotation
    int fore_rel = (int) (prev_fore0 - prev_rear0);
    long long done = cursor0;
    // the inputs of the wavefront's NEXT firing (returns, intensities, pose: one lane per matrix element) are loaded before the current
    // chunk is worked on: a chunk is a load -> ~300 instructions -> barrier chain, and two wavefronts per SIMD cannot hide the load
    float nx_x[RPL], nx_y[RPL], nx_z[RPL];
    uint8_t nx_i[RPL];
    auto load_firing = [&](const long long f)
    {
        const size_t fi = fglob + (size_t) f;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            nx_x[k] = nx_y[k] = nx_z[k] = __builtin_nanf("");
            nx_i[k] = 0;
            if (row < R && f < n)
            {
                const size_t src = (fi * R + row) * 3;
                nx_x[k] = xyz[src];
                nx_y[k] = xyz[src + 1];
                nx_z[k] = xyz[src + 2];
                nx_i[k] = inten[fi * R + row];
            }
        }
    };
    load_firing(cursor0 + wave);
    for (long long f0 = cursor0; f0 < n; f0 += IM_WAVES)
    {
        const long long f = f0 + wave;
        const bool mine = f < n;
        PreppedPoint q[RPL];
        int oc[RPL];
        float cx[RPL], cy[RPL], cz[RPL];
        uint8_t cint[RPL];
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            cx[k] = nx_x[k];
            cy[k] = nx_y[k];
            cz[k] = nx_z[k];
            cint[k] = nx_i[k];
        }
        double T[12]; // (wave-uniform: the matrix travels in SGPRs, by scalar loads — as in k_insert_par)
        {
            const double* Tp = poses + (fglob + (size_t) (mine ? f : cursor0)) * 12;
#pragma unroll
            for (int i = 0; i < 12; i++)
                T[i] = Tp[i];
        }
        load_firing(f + IM_WAVES);
        // ---- prepare this wavefront's firing ------------------------------------------------------------------------------------
        int rear_cir = -1, span = 0;
        if (mine)
        {
            int c0 = -1;
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                q[k].cir = PP_SKIP;
                if (row < R)
                    q[k] = prep_point(cx[k], cy[k], cz[k], T, clockwise, g.az_width);
                const unsigned long long m = __ballot(q[k].cir != PP_SKIP && q[k].cir >= 0 && q[k].cir < NC);
                if (c0 < 0 && m)
                    c0 = __builtin_amdgcn_readlane(q[k].cir, (int) __ffsll((long long) m) - 1);
            }
            if (c0 >= 0)
            {
                // columns relative to the first valid return, unwrapped into (-half, half]; rearmost = minimum, foremost = maximum
                int lo = 0x7fffffff, hi = -0x7fffffff; // (neutral for the negated minimum below)
                bool odd = false; // a return outside [0, NC): leave it to the serial kernel
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    oc[k] = (int) 0x80000000;
                    if (q[k].cir != PP_SKIP)
                    {
                        odd |= q[k].cir < 0 || q[k].cir >= NC;
                        int rel = q[k].cir - c0;
                        rel = rel > half ? rel - NC : (rel < -half ? rel + NC : rel);
                        oc[k] = rel;
                        lo = rel < lo ? rel : lo;
                        hi = rel > hi ? rel : hi;
                    }
                }
                lo = wave_min_i32(lo);
                hi = -wave_min_i32(-hi);
                if (!__any(odd))
                {
                    rear_cir = c0 + lo;
                    rear_cir = rear_cir < 0 ? rear_cir + NC : (rear_cir >= NC ? rear_cir - NC : rear_cir);
                    span = hi - lo;
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                        oc[k] = q[k].cir != PP_SKIP ? oc[k] - lo : -1;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RPL; k++)
            s_col[wave][k * 64 + lane] = (mine && rear_cir >= 0) ? oc[k] : -1;
        if (lane == 0)
        {
            s_rear_cir[wave] = rear_cir;
            s_span[wave] = span;
            if (wave == 0)
                s_stop = IM_WAVES;
        }
        __syncthreads();
        // ---- rear column of every firing of the chunk (every thread the same scalar walk), first firing that ends the run -------------
        int my_rel = 0, my_prev_rel = 0, stop = IM_WAVES;
        {
            int rel = carry_rel, cir = carry_cir, fmax = fore_rel;
            for (int j = 0; j < IM_WAVES; j++)
            {
                if (f0 + j >= n)
                {
                    stop = stop < j ? stop : j;
                    break;
                }
                // (what every thread reads here is the same for all of them: readfirstlane keeps the whole walk on the scalar unit — as vector
                // arithmetic the three walks of a chunk were ~170 of the ~1000 vector instructions a firing costs this kernel)
                const int rc = uniform_i32(s_rear_cir[j]), sp = uniform_i32(s_span[j]);
                const int diff = rc - cir;
                const bool ok = rc >= 0 && ((diff > 0 && diff <= half) || diff < -half) && sp < half; // strictly forward, also across the wrap
                const int delta = ok ? (diff < -half ? diff + NC : diff) : 0;
                const int nrel = rel + delta;
                // taken only while the batch has emitted fewer than limit_columns columns before it (k_insert2's loop head), and while the
                // previous tenant of every ring slot it touches is known to be cleared
                if (!ok || (prev_rear0 + rel) - first_unf0 >= g.limit_columns || prev_rear0 + nrel + sp - RC >= clear_done)
                {
                    stop = stop < j ? stop : j;
                    break;
                }
                if (j == wave)
                {
                    my_rel = nrel;
                    my_prev_rel = rel;
                }
                rel = nrel;
                cir = rc;
                fmax = nrel + sp > fmax ? nrel + sp : fmax;
            }
        }
        // ---- per-row collision rule: this firing's cell of a row must lie ahead of everything the row has written -----------------
        if (mine && wave < stop)
        {
            bool clash = false;
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (oc[k] >= 0)
                {
                    int last = s_rowmax[row];
                    int rel = carry_rel, cir = carry_cir;
                    for (int j = 0; j < wave; j++) // (the rear columns of the earlier firings of the chunk, recomputed: a handful of scalar adds)
                    {
                        const int rc = uniform_i32(s_rear_cir[j]);
                        const int diff = rc - cir;
                        rel += diff < -half ? diff + NC : diff;
                        cir = rc;
                        const int o = s_col[j][row];
                        if (o >= 0)
                            last = rel + o > last ? rel + o : last;
                    }
                    clash |= my_rel + oc[k] <= last;
                }
            }
            if (__any(clash) && lane == 0)
                atomicMin(&s_stop, wave);
        }
        __syncthreads();
        {
            const int st2 = uniform_i32(s_stop);
            stop = st2 < stop ? st2 : stop;
        }
        // ---- accepted firings write their cells ------------------------------------------------------------------------------------
        if (mine && wave < stop)
        {
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (oc[k] >= 0)
                {
                    const int crel = my_rel + oc[k];
                    const unsigned lcq = (unsigned) (lc0 + crel) / (unsigned) RC;
                    const int lc = (int) ((unsigned) (lc0 + crel) - lcq * (unsigned) RC);
                    const unsigned ci = (unsigned) lc * (unsigned) R + (unsigned) row;
                    at32(p.sc_rec, ci) = make_float4(q[k].x, q[k].y, q[k].z, q[k].incl);
                    at32(p.inten, ci) = cint[k];
                    at32(p.src, ci) = (uint32_t) (seq0 + (f - cursor0));
                    at32(p.dist, ci) = q[k].dist;
                    at32(p.incl, ci) = q[k].incl;
                    at32(p.incaz, ci) = q[k].incaz; // (the return's rotation, rot0 + (cir0 + crel) / num_columns, is that of its column)
                    at32(p.gtag, ci) = cell_tag(pass0 + (long long) lcq);
                    atomicMax(&s_rowmax[row], crel);
                }
            }
            // columns [G_(f-1), G_f) (rearmost columns) are finished by this firing and carry its pose (cc.cpp:289-291)
            const int cnt = my_rel - my_prev_rel;
            for (int jj = lane; jj < cnt; jj += 64)
                p.trig[(int) ((unsigned) (lc0 + my_prev_rel + jj) % (unsigned) RC)] = (int) f;
        }
        // ---- carry (the same walk over the accepted firings, in every thread) -------------------------------------------------------
        for (int j = 0; j < stop; j++)
        {
            const int rc = uniform_i32(s_rear_cir[j]), sp = uniform_i32(s_span[j]);
            const int diff = rc - carry_cir;
            carry_rel += diff < -half ? diff + NC : diff;
            carry_cir = rc;
            fore_rel = carry_rel + sp > fore_rel ? carry_rel + sp : fore_rel;
        }
        done = f0 + stop;
        if (stop < IM_WAVES)
            break;
        __syncthreads(); // this chunk's LDS reads and s_rowmax updates are complete before the next chunk rewrites the hand-off arrays
    }
    __syncthreads();
    if (tid == 0)
    {
        st->clear_done = clear_done;
#ifndef CC_A2_STATS
        st->dbg[6] += (unsigned long long) (done - cursor0);
        st->dbg[7] += 1;
#endif
        if (done > cursor0)
        {
            const long long G = prev_rear0 + carry_rel;
            const long long F = prev_rear0 + fore_rel;
            st->prev_rearmost = G;
            st->prev_foremost = F > prev_fore0 ? F : prev_fore0;
            st->first_unfinished = G;
            if (F > ring_end0)
                st->ring_end = F;
            st->cursor = done;
            st->firings_consumed = (unsigned long long) (seq0 + (done - cursor0));
            if (st->pre_seg_begin == 0)
                st->pre_seg_begin = first_unf0;
        }
        if (left_over)
        {
            if (done == n && cursor0 == 0 && done > 0)
            {
                // the whole batch was taken: the columns it finished are [first_unfinished at entry, rearmost column now)
                st->batch[slot].seg_begin = first_unf0;
                st->batch[slot].seg_end = prev_rear0 + carry_rel;
                st->batch[slot].acp_next = first_unf0;
                st->batch[slot].pub_begin = -1;
                st->batch[slot].pub_end = -1;
                st->batch[slot].fused = 0;
            }
            else
                atomicAdd(left_over, 1);
        }
    }
}

// =====================================================================================================
// k_table — sc_inclination_angles_between_lasers_ (cc.cpp:353-357): per row the last non-NaN inclination step over the
// emitted columns, in column order = a per-row "last valid value" scan along the columns. The segmentation needs the table as of every
// column. Round 4: the scan inside a TILE of 64 columns is done where the tile is segmented (k_seg_scan: lanes = columns, one ballot and
// one lane permute per row), so all this kernel leaves is the table as of the column in front of every tile:
//   1  wavefront w walks tiles w, w + TABLE_WAVES, ... (lanes = rows, every column read once): the last valid step INSIDE the tile
//      (NaN: none) -> tabc[tile][row]
//   2  one wavefront, lanes = rows: running "last valid" over the tiles in order, starting from the stream's table; tabc[tile][row]
//      becomes the table in front of the tile, Planes::curtab the table after the batch's last column.
// Streams whose batch went through the fused insertion (BatchDesc::fused, k_insert_par) have their tabc from there.
// grid = streams, block = 64 * TABLE_WAVES.
// =====================================================================================================

// phase 2 (shared with k_insert_par / k_insert_par_fin): tl[t][row] holds the last valid step inside tile t or NaN
template<int RPL>
__device__ __forceinline__ void table_scan_tiles(const SP& p, const int R, const int ntiles, const int lane)
{
    float carry[RPL];
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        const int row = k * 64 + lane;
        carry[k] = row < R ? p.curtab[row] : 0.f;
    }
    constexpr int U = 8;
    for (int t0 = 0; t0 < ntiles; t0 += U)
    {
        float v[U][RPL];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                v[u][k] = (row < R && t0 + u < ntiles) ? p.tabc[(size_t) (t0 + u) * R + row] : __builtin_nanf("");
            }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (row < R && t0 + u < ntiles)
                {
                    p.tabc[(size_t) (t0 + u) * R + row] = carry[k];
                    if (!(v[u][k] != v[u][k]))
                        carry[k] = v[u][k];
                }
            }
    }
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        const int row = k * 64 + lane;
        if (row < R)
            p.curtab[row] = carry[k];
    }
}

template<int RPL>
__global__ __launch_bounds__(64 * TABLE_WAVES) void k_table(Geometry g, Planes P, StreamState* states, int first_stream, int slot)
{
    const int s = first_stream + blockIdx.x;
    const int lane = lane_id(), wave = uniform_i32((int) (threadIdx.x >> 6));
    StreamState* st = &states[s];
#ifdef CC_CHAIN2_PRIO
    __builtin_amdgcn_s_setprio(CC_CHAIN2_PRIO);
#endif
    if (threadIdx.x == 0)
        st->batch[slot].mode = st->assoc_mode; // one decision per batch and stream for every kernel behind this one (any value the
                                               // association chain of the previous batch is just writing is fine)
    const long long seg_begin = st->batch[slot].seg_begin, seg_end = st->batch[slot].seg_end;
    if (seg_begin < 0 || seg_begin >= seg_end)
        return;
    if (st->batch[slot].fused)
        return; // (k_insert_par segmented the batch's per-cell part and left the table carries)
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols;
    const int ntiles = (int) ((seg_end - seg_begin + 63) >> 6);
    constexpr int U = 16;
    for (int t = wave; t < ntiles; t += TABLE_WAVES)
    {
        const long long c_lo = seg_begin + 64ll * t, c_hi = (c_lo + 64 < seg_end ? c_lo + 64 : seg_end);
        float last[RPL]; // NaN = no valid step in this tile so far
#pragma unroll
        for (int k = 0; k < RPL; k++)
            last[k] = __builtin_nanf("");
        int lc = (int) (c_lo % RC);
        for (long long c0 = c_lo; c0 < c_hi; c0 += U)
        {
            float cur[U][RPL], below[U][RPL];
#pragma unroll
            for (int u = 0; u < U; u++)
            {
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    cur[u][k] = below[u][k] = 0.f;
                    if (row < R && c0 + u < c_hi)
                    {
                        const size_t ci = (size_t) lc * R + row;
                        cur[u][k] = p.incl[ci];
                        below[u][k] = row + 1 < R ? p.incl[ci + 1] : 0.f;
                    }
                }
                lc = lc + 1 == RC ? 0 : lc + 1;
            }
#pragma unroll
            for (int u = 0; u < U; u++)
            {
                if (c0 + u >= c_hi)
                    break;
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const float diff = cur[u][k] - below[u][k];
                    if (!(diff != diff))
                        last[k] = diff;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            if (row < R)
                p.tabc[(size_t) t * R + row] = last[k];
        }
    }
    __syncthreads(); // (workgroup-scope release / acquire: the tiles' entries are visible to wavefront 0)
    if (wave == 0)
        table_scan_tiles<RPL>(p, R, ntiles, lane);
}

// ---- k_ego: ego_robot_frame_from_odom_frame = robot_from_sensor * odom_from_sensor^-1 (cc.cpp:300-301) once per FIRING of the batch (one
// thread each) instead of once per column and wavefront, where all 64 lanes evaluated the same ~80 double-precision operations. Same expressions,
// same order. out[(stream in launch * n + firing) * EGO_STRIDE] = {R (3x3, row major), t, skip_r2}. grid = (n / 256, streams).
// skip_r2 (round 4): the ego-box test of cc.cpp:390-403 transforms every return with this matrix in double precision — 18 f64 operations per
// cell to find that a return 20 m away is not on the ego vehicle. With e = M (p - t_T) + A_t (M = A_R R_T^T) a box hit needs |e| < B, B = the
// box's farthest corner, hence sigma_min(M) |p - t_T| - |A_t| < B. skip_r2 is a rigorous upper bound of the squared f32 distance (as the
// segmentation computes it: x2 * x2 + uz * uz, relative to this firing's sensor position) up to which a hit is possible; +inf when the rotation
// blocks are too far from orthonormal to say. Cells beyond it skip the transform; the others evaluate it exactly as before.
__device__ __forceinline__ void ego_record(const StreamState* __restrict__ states, int first_stream, const cc_config& cfg, const double* __restrict__ poses,
                                           long long n, long long n_total, long long fbase, double* __restrict__ out, const int sl, const long long f)
{
    const double* A = states[first_stream + sl].robot_from_sensor;
    const double* T = poses + ((size_t) sl * (size_t) n_total + (size_t) fbase + (size_t) f) * 12;
    double ir[9], it[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            ir[i * 3 + j] = T[j * 4 + i];
    for (int i = 0; i < 3; i++)
        it[i] = ((-ir[i * 3 + 0]) * T[3] + (-ir[i * 3 + 1]) * T[7]) + (-ir[i * 3 + 2]) * T[11];
    double* o = out + ((size_t) sl * (size_t) n + (size_t) f) * EGO_STRIDE;
    for (int i = 0; i < 3; i++)
    {
        for (int j = 0; j < 3; j++)
            o[i * 3 + j] = (A[i * 4 + 0] * ir[0 * 3 + j] + A[i * 4 + 1] * ir[1 * 3 + j]) + A[i * 4 + 2] * ir[2 * 3 + j];
        o[9 + i] = ((A[i * 4 + 0] * it[0] + A[i * 4 + 1] * it[1]) + A[i * 4 + 2] * it[2]) + A[i * 4 + 3];
    }
    // how far the Gram matrix of a 3x3 block is from the identity (Frobenius): sigma_min^2 >= 1 - dev
    auto gram_dev = [](const double* m, const int stride) -> double
    {
        double dev = 0.;
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
            {
                double d = 0.;
                for (int k = 0; k < 3; k++)
                    d += m[k * stride + a] * m[k * stride + b];
                d -= a == b ? 1. : 0.;
                dev += d * d;
            }
        return __builtin_sqrt(dev);
    };
    const double dev_t = gram_dev(T, 4), dev_a = gram_dev(A, 4);
    auto mx2 = [](const float a, const float b) -> double
    {
        const double x = a, y = b;
        return x * x > y * y ? x * x : y * y;
    };
    const double box = __builtin_sqrt(mx2(cfg.length_ref_to_front_end_, cfg.length_ref_to_rear_end_) + mx2(cfg.width_ref_to_left_mirror_, cfg.width_ref_to_right_mirror_) +
                                      mx2(cfg.height_ref_to_maximum_, cfg.height_ref_to_ground_));
    const double at = __builtin_sqrt((A[3] * A[3] + A[7] * A[7]) + A[11] * A[11]);
    double skip = __builtin_inf();
    if (dev_t < 0.5 && dev_a < 0.5 && box == box && at == at) // (NaN anywhere: no skipping)
    {
        const double sigma = __builtin_sqrt((1. - dev_t) * (1. - dev_a));
        const double delta = 2.4e-7 * ((__builtin_fabs(T[3]) + __builtin_fabs(T[7])) + __builtin_fabs(T[11])) + 1e-6;
        const double r = ((box + at) / sigma + delta) * 1.00001;
        const double r2 = r * r * 1.00001;
        float r2f = (float) r2;
        if ((double) r2f < r2)
            r2f = __builtin_bit_cast(float, __builtin_bit_cast(int, r2f) + 1); // round up
        skip = r2f == r2f ? (double) r2f : __builtin_inf();
    }
    o[12] = skip;
}

__global__ __launch_bounds__(256) void k_ego(const StreamState* __restrict__ states, int first_stream, cc_config cfg, const double* __restrict__ poses,
                                             long long n, long long n_total, long long fbase, double* __restrict__ out)
{
    const long long f = (long long) blockIdx.x * 256 + threadIdx.x;
    if (f < n)
        ego_record(states, first_stream, cfg, poses, n, n_total, fbase, out, (int) blockIdx.y, f);
}

// ---- k_seg_pre: the per-cell part for columns whose cells come from the ring (everything the fused insertion did not take). Lanes = rows
// (coalesced); one wavefront per chunk of consecutive columns. grid = (streams, SEGPRE_BLOCKS), block = 64.

template<int RPL>
__global__ __launch_bounds__(64) void k_seg_pre(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot,
                                                const double* __restrict__ poses, long long n_total, long long fbase,
                                                const double* __restrict__ ego, long long n_batch)
{
    const int sl = blockIdx.x;
    const int s = first_stream + sl;
    StreamState* st = &states[s];
    const long long seg_begin = st->batch[slot].seg_begin, seg_end = st->batch[slot].seg_end;
    if (seg_begin < 0 || seg_begin >= seg_end)
        return;
    if (st->batch[slot].fused)
        return;
    if (!st->has_robot_tf)
    {
        if (blockIdx.y == 0 && lane_id() == 0)
            raise_error(st, CC_ERR_NO_ROBOT_TRANSFORM, seg_begin, 0);
        return;
    }
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols;
    const int lane = lane_id();

    // this wavefront's chunk of the batch's columns
    const long long total = seg_end - seg_begin;
    const long long chunk_len = (total + SEGPRE_BLOCKS - 1) / SEGPRE_BLOCKS;
    const long long c_lo = seg_begin + chunk_len * (long long) blockIdx.y, c_hi = (c_lo + chunk_len < seg_end ? c_lo + chunk_len : seg_end);
    if (c_lo >= c_hi)
        return;
    // (ring column, rotation index and ring pass advanced incrementally: a 64-bit division per column costs ~100 scalar instructions)
    const int NC = g.num_columns;
    int lc = (int) (c_lo % RC);
    long long rot = c_lo / NC;
    int cir = (int) (c_lo - rot * NC);
    long long pass = c_lo / RC; // pass over the ring (cell_tag)
    // The ring-pass tags (which say which cells hold a record at all) are loaded one column ahead, the cells at the top of their column.
    // (Loading the cells a column ahead as well cost a second set of cell registers — 87 instead of 79 VGPRs — and with them more
    // occupancy than the read-ahead hid: − 2 % on the step at 64 rows, − 4 % at 128.)
    uint16_t a_tg[RPL];            // tags of column gc + 1
    uint16_t n_tg[RPL];            // column gc's tags ...
    float n_dist[RPL], n_incaz[RPL];
    float4 n_rec[RPL];             // ... and cells
    uint8_t n_inten[RPL];
    int n_trig = 0;
    auto load_tags = [&](const long long gcx, const int lcx)
    {
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            a_tg[k] = CELL_CLEARED;
            if (row < R && gcx < c_hi)
                a_tg[k] = p.gtag[(size_t) lcx * R + row];
        }
    };
    auto load_cells = [&](const long long gcx, const int lcx, const uint16_t tagx)
    {
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            n_dist[k] = n_incaz[k] = 0.f;
            n_inten[k] = 0;
            // a cell that received a return carries its record; a cleared cell has inclination = NaN (cc.cpp:1110-1119) and nothing else
            n_rec[k] = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            if (row < R && gcx < c_hi)
            {
                const size_t ci = (size_t) lcx * R + row;
                n_dist[k] = p.dist[ci];
                if (n_tg[k] == tagx)
                {
                    n_rec[k] = p.sc_rec[ci];
                    n_incaz[k] = p.incaz[ci];
                    n_inten[k] = p.inten[ci];
                }
            }
        }
        if (gcx < c_hi)
            n_trig = p.trig[lcx];
    };
    load_tags(c_lo, lc);
    CazBase cb = caz_base_of_rotation(rot); // (recomputed where the rotation changes: two f64 products and two 64-bit conversions)
    long long cb_rot = rot;
    for (long long gc = c_lo; gc < c_hi; gc++, pass += (lc + 1 == RC ? 1 : 0), lc = (lc + 1 == RC ? 0 : lc + 1), rot += (cir + 1 == NC ? 1 : 0),
                   cir = (cir + 1 == NC ? 0 : cir + 1))
    {
        const size_t base = (size_t) lc * R;
        if (rot != cb_rot)
        {
            cb = caz_base_of_rotation(rot);
            cb_rot = rot;
        }
        const uint16_t tag = cell_tag(pass);
        // this column's cells (its tags arrived during the previous column), then the tags of the next one
        {
#pragma unroll
            for (int k = 0; k < RPL; k++)
                n_tg[k] = a_tg[k];
            load_cells(gc, lc, tag);
            load_tags(gc + 1, lc + 1 == RC ? 0 : lc + 1);
        }
        // the caller's [stream][n_total] pose buffer; this batch is its firings [fbase, ...), trig is relative to the batch
        const int trig = uniform_i32(n_trig); // (wave-uniform: the pose and the matrices below arrive by scalar loads)
        const double* T = poses + ((size_t) sl * (size_t) n_total + (size_t) fbase + (size_t) trig) * 12;
        // ego_robot_frame_from_odom_frame = robot_from_sensor * odom_from_sensor^-1   (cc.cpp:300-301), prepared per firing by k_ego
        const double* E = ego + ((size_t) sl * (size_t) n_batch + (size_t) trig) * EGO_STRIDE;
        const float spx = (float) T[3], spy = (float) T[7], spz = (float) T[11]; // sgps_sensor_position (cc.cpp:111-113)

        float cx[RPL], cy[RPL], cz[RPL], dist[RPL], incl[RPL];
        bool empty_cell[RPL], overrun = false;
        int overrun_row = -1;        // the reference walks the rows bottom-up and reports the first stale cell it meets (cc.cpp:314-345)
        long long overrun_gcol = -1;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            dist[k] = incl[k] = __builtin_nanf("");
            cx[k] = cy[k] = cz[k] = 0.f;
            empty_cell[k] = false;
            if (row < R)
            {
                const uint16_t tg = n_tg[k];
                if (tg != tag && tg != CELL_CLEARED)
                {
                    overrun = true; // cc.cpp:320-345
                    overrun_row = row;
                    // the stale global column index: this ring column in the latest earlier pass that carries the cell's tag
                    overrun_gcol = gc - (long long) ((((unsigned) tag - (unsigned) tg) & 0x7fffu)) * RC;
                }
                empty_cell[k] = tg != tag;
                dist[k] = n_dist[k];
                cx[k] = n_rec[k].x;
                cy[k] = n_rec[k].y;
                cz[k] = n_rec[k].z;
                incl[k] = n_rec[k].w;
            }
        }
        if (__any(overrun))
        {
            // Columns are segmented in parallel here; the reference meets the lowest stale column first. Keep the minimum; the host
            // fills in error_a / error_b from that column's cells (cc_engine.hip: fixup_overrun).
            const int worst = -wave_min_i32(-overrun_row); // highest stale row = the first one of the reference's bottom-up walk
            if (overrun_row == worst)
            {
                atomicMin((unsigned long long*) &st->overrun_col, (unsigned long long) gc);
                raise_error(st, CC_ERR_RING_OVERRUN, overrun_gcol, gc);
            }
            continue;
        }
        float x2[RPL], uz[RPL], w[RPL];
        int flags[RPL];
        seg_pre_cells<RPL>(cfg, R, lane, cx, cy, cz, dist, incl, n_inten, spx, spy, spz, E, x2, uz, w, flags);
        int kpos = 0x7fffffff, kneg = 0x7fffffff;
        bool any_empty = false;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            if (row >= R)
                continue;
            const size_t ci = base + row;
            if (empty_cell[k])
                p.gtag[ci] = tag; // cells that received a return already carry it (insertion kernels)
            // (continuous azimuth of a cell without a return: cc.cpp:371-372 — not stored: every reader knows the cell's column)
            if (flags[k] & SG_NAN)
                any_empty = true;
            else
                caz_key(n_incaz[k], kpos, kneg);
            p.sg_x2[ci] = x2[k];
            p.sg_uz[ci] = uz[k];
            p.sg_w[ci] = w[k];
            p.sg_flags[ci] = (uint8_t) flags[k];
        }
        const double min_az = column_min_caz(cb, kpos, kneg, any_empty, gc, g.az_width);
        if (lane == 0)
        {
            p.colg[lc] = gc;
            p.colminaz[lc] = min_az;
        }
    }
}

// ---- k_seg_scan: the part of the segmentation that runs along the rows of a column (cc.cpp:306-565 state machine + downward fix-up +
// ignore flags 567-616) and, since round 4, everything that needs sc_inclination_angles_between_lasers_ (cc.cpp:353-357): the table as of
// every column, the supplemented inclination of cells without a return (:364-369) and the inclination-step filter (:597-603) of the cells
// whose own column has no valid step.
// One lane per column on tiles of 64 columns; grid = (streams, tiles of 64 columns), block = 64, dynamic LDS = seg_scan_lds_bytes(num_rows).
// The table along the columns of a tile: a lane whose cell has a valid step to the row below holds it (staging plane sg_w); the table entry of
// row r as of column c is the step of the nearest such lane at or before c — one ballot, one count-leading-zeros and one lane permute per row —
// or, when the tile has none before c, the table in front of the tile (Planes::tabc: k_table / k_insert_par).
// The staged inputs are column-major like every plane of the ring, so a lane that read its own
// column touched a different 128-byte line than its neighbours with every load, 32 bytes at a time: round 2 measured 1.42 GB fetched per step for
// 0.32 GB of input (the lines did not survive in L2 next to the other chains). Round 3: the wavefront loads 16 rows x 64 columns at a time with
// lanes = (column, 16-byte piece) — 64 contiguous bytes per column and plane, every line fetched once —, hands them to the column lanes through LDS
// (XOR-swizzled 16-byte pieces: conflict-free both ways) one chunk ahead of the scan, and the flags of the whole tile start out in the output tile
// (a cell's flag byte is replaced by its result when its row is done).
// The look-back of the state machine (cc.cpp:513-535) walks down from a new obstacle over the ground cells right below it: rarely more than a few
// rows. The tile keeps the azimuth-plane distance of two chunks (the current one and the one below) and reads deeper rows from the staging plane.
// Row counts that are not a multiple of 16 take the round-2 form (every lane reads its own column, 8 rows at a time; 16 rows of look-back in LDS).
constexpr int SEG_X2_RING = 16;
constexpr int SEG_CH = 16; // rows per chunk of the tiled form
constexpr int SEG_FEW = 4; // tiles of at most this many columns are loaded whole (3 * SEG_FEW * rows floats fit the chunk buffers up to 341 rows)
__host__ __device__ inline int seg_pitch_f(int R)
{
    (void) R;
    return SEG_X2_RING + 1; // odd number of words per column
}
__host__ __device__ inline int seg_pitch_b(int R)
{
    return ((R + 3) & ~3) + 4; // bytes per column: multiple of 4 whose word count is odd
}
__host__ __device__ inline bool seg_tiled(int R)
{
    return R >= SEG_CH && (R % SEG_CH) == 0;
}
__host__ inline size_t seg_scan_lds_bytes(int R)
{
    const size_t f = seg_tiled(R) ? (size_t) 4 * 64 * SEG_CH * 4 : (size_t) 64 * seg_pitch_f(R) * 4;
    return f + (size_t) 64 * seg_pitch_b(R);
}

// compact codes of the label values inside the LDS tile
enum
{
    SG_G_UNKNOWN = 0, SG_G_GROUND = 1, SG_G_OBSTACLE = 2, SG_G_EGO = 3, SG_G_FOG = 4,
    SG_D_WHITE = 0, SG_D_GRAY = 1, SG_D_ORANGE = 2, SG_D_GREEN = 3, SG_D_YELLOWGREEN = 4, SG_D_YELLOW = 5, SG_D_RED = 6, SG_D_DARKRED = 7,
    SG_D_VIOLET = 8, SG_D_LIGHTGRAY = 9
};

// (20 KB of LDS per wavefront: two of them per SIMD at most — the register budget that goes with that, not 128)
__global__ __launch_bounds__(64, 2) void k_seg_scan(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot)
{
    const int s = first_stream + blockIdx.x;
    StreamState* st = &states[s];
    const long long seg_begin = st->batch[slot].seg_begin, seg_end = st->batch[slot].seg_end;
    if (seg_begin < 0 || st->error != 0)
        return;
    const long long tile0 = seg_begin + (long long) blockIdx.y * 64;
    if (tile0 >= seg_end)
        return;
    const int ncols = (int) (seg_end - tile0 < 64 ? seg_end - tile0 : 64);
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols;
    const int lane = lane_id();
    const int PF = seg_pitch_f(R), PB = seg_pitch_b(R);
    const bool tiled = seg_tiled(R);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // The tile keeps only what the state machine looks back at: the azimuth-plane distance of the rows below (cc.cpp:513-535) and
    // one output byte per cell (bits 0-2 ground label code, bits 3-6 debug label code, bit 7 "ignored if it ends up an obstacle").
    float* l_x2 = (float*) smem;
    unsigned char* l_out = (unsigned char*) (l_x2 + (tiled ? 4 * 64 * SEG_CH : 64 * PF));

    const int lc0 = (int) (tile0 % RC);
    if (!(g.debug_flags & 1))
    {
        const bool active = lane < ncols;
        const long long gc = tile0 + lane;
        int lcl = lc0 + lane;
        lcl = lcl >= RC ? lcl - RC : lcl;
        const float* gx = p.sg_x2 + (size_t) lcl * R;
        const float* gz = p.sg_uz + (size_t) lcl * R;
        const float* gw = p.sg_w + (size_t) lcl * R;
        const unsigned char* gf = p.sg_flags + (size_t) lcl * R;
        float4* g_rec = p.sc_rec + (size_t) lcl * R; // (cells without a return: {NaN, NaN, NaN, supplemented inclination}, what the window scan reads)
        float* g_incl = p.incl + (size_t) lcl * R;
        const float* tab_in = p.tabc + (size_t) blockIdx.y * R; // the table in front of this tile (wave-uniform: scalar loads)
        unsigned char* oo = l_out + lane * PB;
        const float height_sensor_to_ground = -(float) st->robot_from_sensor[11] + cfg.height_ref_to_ground_;
        const bool chess_odd = cfg.ignore_points_in_chessboard_pattern && (gc & 1); // column parity (cc.cpp:600-606)
        const bool chess_even = cfg.ignore_points_in_chessboard_pattern && !(gc & 1);
        bool first_obstacle_detected = false, first_point_found = false;
        float lg2x = 0.f, lgz = height_sensor_to_ground; // last (quite certain) ground point in the azimuth plane
        float pv2x = 0.f, pvz = 0.f;
        unsigned char previous_label = 0;
        // one row of the state machine: f = the cell's flags (k_seg_pre), (cur2x, cur2y) = the point in the azimuth plane;
        // x2_below(row) = the azimuth-plane distance of a row below.
        // Round 4: written WITHOUT divergent branches. As nested ifs the compiler turned a row into 22 s_and_saveexec / s_cbranch_execz pairs and
        // ~90 scalar mask operations — on a lone wavefront every one of those branches costs 15 - 30 clocks (lone-wave cost model) —
        // so every quantity is computed for every lane (garbage where the cell has no return: nothing traps) and the cases are selects. The one
        // loop (the downward fix-up of cc.cpp:513-535) stays a loop behind a wave-uniform test.
        auto row_step = [&](const int row, const int f, const float cur2x, const float cur2y, auto&& x2_below)
        {
            const bool valid = (f & (SG_NAN | SG_FOG | SG_EGO)) == 0;
            const bool first = valid & !first_point_found;
            const bool normal = valid & first_point_found;
            // cc.cpp:567-616 for a point that ends up an obstacle: too close / inclination filter / chessboard thinning
            const bool ign = ((f & (SG_TOO_CLOSE | SG_INCL_IGNORE)) != 0) | ((row & 1) ? chess_even : chess_odd);
            // the first point outside the ego box (cc.cpp:408-432)
            const float h = cur2y - height_sensor_to_ground;
            const bool first_ground = (h > cfg.first_ring_as_ground_min_allowed_z_diff) & (h < cfg.first_ring_as_ground_max_allowed_z_diff);
            // slopes w.r.t. the previous point and the last certain ground point (cc.cpp:434-447)
            const float p2cx = cur2x - pv2x, p2cy = cur2y - pvz;
            const float slope_to_prev = p2cy / p2cx;
            const bool flat_prev = (ccm::absf(slope_to_prev) < cfg.max_slope) & (p2cx > 0) & ((cfg.use_terrain == 0) | (p2cx < 5));
            const float l2cx = cur2x - lg2x, l2cy = cur2y - lgz;
            const float slope_to_lg = l2cy / l2cx;
            const bool flat_lg = (ccm::absf(slope_to_lg) < cfg.max_slope) & (l2cx > 0);
            const bool no_terrain = cfg.use_terrain == 0;
            const bool green = !first_obstacle_detected & flat_prev;                                                     // cc.cpp:450-454
            const bool yellowgreen = !green & no_terrain & first_obstacle_detected & flat_prev & flat_lg;                // :489-493
            const bool yellow = !green & !yellowgreen & no_terrain &
                                (ccm::absf(l2cx) < cfg.ground_because_close_to_last_certain_ground_max_dist_diff) &
                                (ccm::absf(l2cy) < cfg.ground_because_close_to_last_certain_ground_max_z_diff);           // :494-500
            const bool ground_n = green | yellowgreen | yellow;
            const unsigned d_n = green ? (unsigned) SG_D_GREEN : (yellowgreen ? (unsigned) SG_D_YELLOWGREEN : (yellow ? (unsigned) SG_D_YELLOW : (unsigned) SG_D_RED));
            const unsigned g_n = ground_n ? (unsigned) SG_G_GROUND : (unsigned) SG_G_OBSTACLE;
            const unsigned d_f = first_ground ? (unsigned) SG_D_GRAY : (unsigned) SG_D_ORANGE;
            const unsigned g_f = first_ground ? (unsigned) SG_G_GROUND : (unsigned) SG_G_OBSTACLE;
            unsigned g = first ? g_f : g_n, d = first ? d_f : d_n;
            g = (f & SG_EGO) ? (unsigned) SG_G_EGO : g;
            d = (f & SG_EGO) ? (unsigned) SG_D_VIOLET : d;
            g = (f & SG_FOG) ? (unsigned) SG_G_FOG : g;
            d = (f & SG_FOG) ? (unsigned) SG_D_LIGHTGRAY : d;
            g = (f & SG_NAN) ? (unsigned) SG_G_UNKNOWN : g;
            d = (f & SG_NAN) ? (unsigned) SG_D_WHITE : d;
            const bool red = normal & !ground_n;
            if (__any(red))
            {
                // cc.cpp:513-535: go down in the rows and mark very close (ground) points as obstacle too — nearly always over after one look
                int below = row + 1;
                bool go = red & (below < R);
                while (__any(go))
                {
                    const int bi = go ? below : row + 1 < R ? row + 1 : row; // (lanes that are through look at a harmless row)
                    const unsigned bo = oo[bi];
                    const unsigned bg = bo & 7u, bd = (bo >> 3) & 15u;
                    const float xb = x2_below(bi, go);
                    const bool is_ground = bg == (unsigned) SG_G_GROUND;
                    const bool cont = go & ((bd == (unsigned) SG_D_YELLOW) |
                                            (is_ground & (ccm::absf(cur2x - xb) < cfg.obstacle_because_next_certain_obstacle_max_dist_diff)));
                    if (cont & is_ground)
                        oo[bi] = (unsigned char) ((bo & 0x80u) | SG_G_OBSTACLE | (SG_D_DARKRED << 3));
                    below += cont ? 1 : 0;
                    go = cont & (below < R);
                }
            }
            // check whether we have ever seen an obstacle; the last (certain) ground point (cc.cpp:538-560)
            first_obstacle_detected = first ? !first_ground : (first_obstacle_detected | red);
            const bool keep_as_ground = normal & (green | yellowgreen) & (slope_to_prev > cfg.last_ground_point_slope_higher_than) &
                                        (ccm::absf(p2cx) < cfg.last_ground_point_distance_smaller_than) & (previous_label != SG_D_YELLOW);
            const bool new_lg = (first & first_ground) | keep_as_ground;
            lg2x = new_lg ? cur2x : lg2x;
            lgz = new_lg ? cur2y : lgz;
            pv2x = valid ? cur2x : pv2x;
            pvz = valid ? cur2y : pvz;
            previous_label = valid ? (unsigned char) d : previous_label;
            first_point_found |= valid;
            oo[row] = (unsigned char) (g | (d << 3) | ((valid & ign) ? 0x80u : 0u));
        };
        // ---- the table along the columns, the supplemented inclination and the pending inclination-step tests of one row, then its state machine
        // step. EVERY lane comes here for every row (lanes beyond the tile as columns without returns): the ballot and the permute are wave-wide.
        const unsigned long long le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1ull); // lanes at or before this one
        const bool supplement = cfg.supplement_inclination_angle_for_nan_cells != 0;
        const bool step_filter = cfg.ignore_points_with_too_big_inclination_angle_diff != 0;
        float supp_below = __builtin_nanf(""); // inclination the row below ended up with, if it had no return
        bool below_nan = false;
        // `stash(tab)` is called (predicated, no branch around it) by lanes whose test survives both bounds: the exact evaluation — ~100 instructions,
        // ~1 % of the far cells — is done behind the chunk's rows (the row loops are unrolled: one copy of it per loop, not sixteen); returns "pending"
        auto row_all = [&](const int row, const int f, const float cur2x, const float cur2y, const float wv, const float carry, auto&& x2_below,
                           auto&& stash) -> bool
        {
            const bool own = !(f & (SG_NAN | SG_PENDING)); // this cell's step to the row below is valid: it IS the table entry as of this column
            const unsigned long long m = __ballot(own) & le_mask;
            const int src = m ? 63 - __clzll((long long) m) : lane;
            const float got = __shfl(wv, src, 64);
            const float tab = m ? got : carry; // sc_inclination_angles_between_lasers_[row] after this column (cc.cpp:353-357)
            // cc.cpp:364-369: the inclination of the cell below (after ITS supplement) + the table entry. (Without the option, and in the last row,
            // the cell keeps the inclination of a cell without a return: NaN. Branch-free like row_step.)
            const bool is_nan = (f & SG_NAN) != 0;
            const float supp = (supplement & (row < R - 1)) ? (below_nan ? supp_below : wv) + tab : __builtin_nanf("");
            if (is_nan & active)
            {
                g_rec[row] = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), supp);
                g_incl[row] = supp;
            }
            supp_below = is_nan ? supp : supp_below;
            below_nan = is_nan;
            // cc.cpp:597-603 with the table entry of an earlier column: atan2f(max_distance, distance) < tab. Two rigorous bounds first
            // (seg_pre_cells has the first; the second: atan2f(y, x) <= (y / x) (1 + 3 * 2^-23) for positive arguments)
            const bool pend = ((f & (SG_PENDING | SG_NAN)) == SG_PENDING) & step_filter & (row < R - 1) & !(tab != tab);
            const float a = wv * tab; // (wv: the distance of a pending cell)
            const bool in_range = (cfg.max_distance > 0.f) & (tab >= 0.f) & (tab < 0.05f) & (wv > 0.f) & (wv < 3.0e38f);
            const bool surely_not = in_range & (cfg.max_distance >= 1.01f * a);
            const bool surely = in_range & !surely_not & (cfg.max_distance * 1.000002f < a);
            const int fx = f | ((pend & surely) ? SG_INCL_IGNORE : 0);
            const bool need = pend & !surely_not & !surely;
            if (need)
                stash(tab);
            row_step(row, fx, cur2x, cur2y, x2_below);
            return need;
        };
        if (tiled && ncols <= SEG_FEW && 3 * SEG_FEW * R <= 4 * 64 * SEG_CH) // (the whole columns of three planes fit the chunk buffers)
        {
            // ---- a tile of a few columns (calls of a few firings: the per-column latency path, and the last tile of a batch): the whole columns are
            // loaded with lanes = rows in ONE round trip (the chunked form below spends four dependent ones, 2 us each, on a tile whose scan takes 3 us),
            // then lane c scans column c out of LDS
            float* cx2 = l_x2;
            float* cuz = l_x2 + SEG_FEW * R;
            float* cw = l_x2 + 2 * SEG_FEW * R;
            for (int c = 0; c < ncols; c++)
            {
                int l = lc0 + c;
                l = l >= RC ? l - RC : l;
                for (int row = lane; row < R; row += 64)
                {
                    cx2[c * R + row] = p.sg_x2[(size_t) l * R + row];
                    cuz[c * R + row] = p.sg_uz[(size_t) l * R + row];
                    cw[c * R + row] = p.sg_w[(size_t) l * R + row];
                    l_out[c * PB + row] = p.sg_flags[(size_t) l * R + row];
                }
            }
            wave_lds_fence();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (lane c reads what all lanes wrote)
            {
                const int lc_ = active ? lane : 0; // (lanes beyond the tile read column 0's floats and take them for a column without returns)
                const float* mx = cx2 + lc_ * R;
                const float* mz = cuz + lc_ * R;
                const float* mw = cw + lc_ * R;
                auto x2_below = [&](const int below, const bool wanted) -> float { (void) wanted; return mx[below]; };
                // the table in front of the tile, one row per lane (read back with v_readlane: a scalar load per row would drain the LDS counter)
                const float carry_lo = lane < R ? tab_in[lane] : 0.f, carry_hi = 64 + lane < R ? tab_in[64 + lane] : 0.f;
                auto carry_of = [&](const int row) -> float
                {
                    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, row < 64 ? carry_lo : carry_hi), row & 63));
                };
                for (int b = R - 4; b >= 0; b -= 4)
                {
                    const float4 a = *(const float4*) (mx + b);
                    const float4 c4 = *(const float4*) (mz + b);
                    const float4 w4 = *(const float4*) (mw + b);
                    const float x4[4] = {a.x, a.y, a.z, a.w}, z4[4] = {c4.x, c4.y, c4.z, c4.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
                    const unsigned fw = active ? *(const unsigned*) (oo + b) : 0x01010101u * (unsigned) SG_NAN;
                    unsigned pend = 0;
#pragma unroll
                    for (int u = 3; u >= 0; u--)
                        if (row_all(b + u, (int) ((fw >> (8 * u)) & 0xffu), x4[u], z4[u], ww[u], carry_of(b + u), x2_below,
                                    [&](const float tab) { cuz[lc_ * R + b + u] = tab; })) // (the row's height has been consumed: its slot takes the table entry)
                            pend |= 1u << u;
                    if (__any(pend != 0))
                        for (int u = 0; u < 4; u++)
                            if (((pend >> u) & 1) && ccm::atan2f_exact(cfg.max_distance, mw[b + u]) < mz[b + u])
                                oo[b + u] |= 0x80; // (bit 7 only matters for a cell that ends up an obstacle, and nothing in the state machine reads it)
                }
            }
        }
        else if (tiled)
        {
            // ---- tiled form: lanes = (column of a group of 16, 16-byte piece) while loading, lanes = columns while scanning
            float* t_uz = l_x2 + 2 * 64 * SEG_CH; // l_x2: two chunks (index (row / 16) & 1), t_uz / t_w: the current one
            float* t_w = l_x2 + 3 * 64 * SEG_CH;
            const int ld_c = lane >> 2, ld_q = lane & 3;
            int ld_off[4]; // cell index of row 0 of this lane's four load columns (-1: beyond the tile)
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const int c = j * 16 + ld_c;
                int l = lc0 + c;
                l = l >= RC ? l - RC : l;
                ld_off[j] = c < ncols ? l * R : -1;
            }
            float4 nx[4], nz[4], nw[4];
            float ncarry = 0.f; // the table in front of the tile for the chunk's 16 rows, one per lane (read back with v_readlane)
            auto load_chunk = [&](const int b)
            {
                ncarry = tab_in[b + (lane & 15)];
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    nx[j] = nz[j] = nw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ld_off[j] >= 0)
                    {
                        nx[j] = *(const float4*) (p.sg_x2 + (size_t) ld_off[j] + b + ld_q * 4);
                        nz[j] = *(const float4*) (p.sg_uz + (size_t) ld_off[j] + b + ld_q * 4);
                        nw[j] = *(const float4*) (p.sg_w + (size_t) ld_off[j] + b + ld_q * 4);
                    }
                }
            };
            int b = R - SEG_CH;
            load_chunk(b);
            // flags of the whole tile -> output tile (16 bytes per lane and pass: the pieces of a column are neighbours). Four passes' loads are in
            // flight together (a one-column call is a chain of dependent round trips otherwise: 2 us each)
            {
                const int npieces = R >> 4;
                for (int idx0 = lane; idx0 < 64 * npieces; idx0 += 4 * 64)
                {
                    uint4 v[4];
                    int dst[4];
#pragma unroll
                    for (int u = 0; u < 4; u++)
                    {
                        const int idx = idx0 + u * 64;
                        const int c = idx / npieces, piece = idx - c * npieces;
                        dst[u] = -1;
                        v[u] = make_uint4(0, 0, 0, 0);
                        if (idx < 64 * npieces && c < ncols)
                        {
                            int l = lc0 + c;
                            l = l >= RC ? l - RC : l;
                            v[u] = *(const uint4*) (p.sg_flags + (size_t) l * R + piece * 16);
                            dst[u] = c * PB + piece * 16;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (dst[u] >= 0)
                        {
                            unsigned* d = (unsigned*) (l_out + dst[u]);
                            d[0] = v[u].x, d[1] = v[u].y, d[2] = v[u].z, d[3] = v[u].w;
                        }
                }
            }
            // 16-byte piece q of column c inside a chunk buffer (floats): XOR swizzle, conflict-free for both lane mappings
            auto piece_at = [](const int c, const int q) { return (c * 4 + (q ^ ((c >> 2) & 3))) * 4; };
            for (; b >= 0; b -= SEG_CH)
            {
                float* cx = l_x2 + ((b >> 4) & 1) * (64 * SEG_CH);
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    const int c = j * 16 + ld_c;
                    *(float4*) (cx + piece_at(c, ld_q)) = nx[j];
                    *(float4*) (t_uz + piece_at(c, ld_q)) = nz[j];
                    *(float4*) (t_w + piece_at(c, ld_q)) = nw[j];
                }
                const int carry_bits = __builtin_bit_cast(int, ncarry);
                if (b >= SEG_CH)
                    load_chunk(b - SEG_CH);
                wave_lds_fence(); // one wavefront per block: its LDS accesses execute in order
                {
                    auto x2_below = [&](const int below, const bool wanted) -> float
                    {
                        // this chunk or the one below it: LDS; deeper: the staging plane (the LDS word is read either way: a select between
                        // an LDS and a global address would make this a flat access)
                        float v = l_x2[((below >> 4) & 1) * (64 * SEG_CH) + piece_at(lane, (below & 15) >> 2) + (below & 3)];
                        const bool deep = wanted & (below >= b + 2 * SEG_CH);
                        if (__any(deep))
                            if (deep)
                                v = gx[below];
                        return v;
                    };
                    // four rows (one 16-byte piece per plane) per iteration of a ROLLED loop: the state machine's code stays small
                    // (sixteen unrolled copies of it, for three forms of this kernel, were 25 k instructions)
                    unsigned pend = 0;
#pragma unroll 1
                    for (int q = 3; q >= 0; q--)
                    {
                        const int at = piece_at(lane, q);
                        const float4 a = *(const float4*) (cx + at);
                        const float4 c4 = *(const float4*) (t_uz + at);
                        const float4 w4 = *(const float4*) (t_w + at);
                        const float x4[4] = {a.x, a.y, a.z, a.w}, z4[4] = {c4.x, c4.y, c4.z, c4.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
                        const unsigned fw = active ? *(const unsigned*) (oo + b + q * 4) : 0x01010101u * (unsigned) SG_NAN;
#pragma unroll
                        for (int u = 3; u >= 0; u--)
                            if (row_all(b + q * 4 + u, (int) ((fw >> (8 * u)) & 0xffu), x4[u], z4[u], ww[u],
                                        __builtin_bit_cast(float, __builtin_amdgcn_readlane(carry_bits, q * 4 + u)), x2_below,
                                        [&](const float tab) { t_uz[at + u] = tab; })) // (the row's height is in registers: its slot takes the table entry)
                                pend |= 1u << (q * 4 + u);
                    }
                    if (__any(pend != 0))
                        for (int u = 0; u < SEG_CH; u++)
                        {
                            const int at = piece_at(lane, u >> 2) + (u & 3);
                            if (((pend >> u) & 1) && ccm::atan2f_exact(cfg.max_distance, t_w[at]) < t_uz[at])
                                oo[b + u] |= 0x80; // (bit 7 only matters for a cell that ends up an obstacle, and nothing in the state machine reads it)
                        }
                }
                wave_lds_fence(); // (the next chunk's pieces are stored behind this chunk's reads)
            }
        }
        else
        {
            // ---- rows not a multiple of 16: the inputs are read by the lane that consumes them, 8 rows (one 32-byte sector per plane) at a time
            // and one chunk ahead
            float* x2 = l_x2 + lane * PF;
            const bool vec = (R & 7) == 0; // rows come in whole, aligned 32-byte sectors
            float nx[8], nz[8], nw[8];
            unsigned nf0 = 0, nf1 = 0; // flags of the 8 rows, one byte each
            auto load_chunk = [&](int b) // rows b .. b + 7 (b may be negative in the last chunk of an odd-sized column)
            {
                if (!active)
                {
                    nf0 = nf1 = 0x01010101u * (unsigned) SG_NAN;
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        nx[u] = nz[u] = nw[u] = 0.f;
                }
                else if (vec)
                {
                    const float4 a0 = *(const float4*) (gx + b), a1 = *(const float4*) (gx + b + 4);
                    const float4 c0 = *(const float4*) (gz + b), c1 = *(const float4*) (gz + b + 4);
                    const float4 w0 = *(const float4*) (gw + b), w1 = *(const float4*) (gw + b + 4);
                    const uint2 ff = *(const uint2*) (gf + b);
                    nx[0] = a0.x, nx[1] = a0.y, nx[2] = a0.z, nx[3] = a0.w, nx[4] = a1.x, nx[5] = a1.y, nx[6] = a1.z, nx[7] = a1.w;
                    nz[0] = c0.x, nz[1] = c0.y, nz[2] = c0.z, nz[3] = c0.w, nz[4] = c1.x, nz[5] = c1.y, nz[6] = c1.z, nz[7] = c1.w;
                    nw[0] = w0.x, nw[1] = w0.y, nw[2] = w0.z, nw[3] = w0.w, nw[4] = w1.x, nw[5] = w1.y, nw[6] = w1.z, nw[7] = w1.w;
                    nf0 = ff.x;
                    nf1 = ff.y;
                }
                else
                {
                    nf0 = nf1 = 0;
#pragma unroll
                    for (int u = 0; u < 8; u++)
                    {
                        const int rr = b + u;
                        nx[u] = rr >= 0 ? gx[rr] : 0.f;
                        nz[u] = rr >= 0 ? gz[rr] : 0.f;
                        nw[u] = rr >= 0 ? gw[rr] : 0.f;
                        const unsigned f = rr >= 0 ? gf[rr] : (unsigned) SG_NAN;
                        if (u < 4)
                            nf0 |= f << (8 * u);
                        else
                            nf1 |= f << (8 * (u - 4));
                    }
                }
            };
            int b = R - 8; // lowest row of the chunk being processed; chunks run from the bottom ring (row R - 1) upwards
            load_chunk(b);
            for (; b > -8; b -= 8)
            {
                float x8[8], z8[8], w8[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                {
                    x8[u] = nx[u];
                    z8[u] = nz[u];
                    w8[u] = nw[u];
                }
                const unsigned f0 = nf0, f1 = nf1;
                if (b - 8 > -8)
                    load_chunk(b - 8);
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (b + u >= 0)
                        x2[(b + u) & (SEG_X2_RING - 1)] = x8[u];
                auto x2_below = [&](const int below, const bool wanted) -> float
                {
                    float v = x2[below & (SEG_X2_RING - 1)];
                    const bool deep = wanted & (below >= b + SEG_X2_RING);
                    if (__any(deep))
                        if (deep)
                            v = gx[below];
                    return v;
                };
#pragma unroll
                for (int u = 7; u >= 0; u--)
                {
                    const int row = b + u;
                    if (row < 0)
                        break;
                    float tab_u = 0.f;
                    const bool pend = row_all(row, (int) (((u < 4 ? f0 : f1) >> (8 * (u & 3))) & 0xffu), x8[u], z8[u], w8[u], tab_in[row], x2_below,
                                              [&](const float tab) { tab_u = tab; });
                    if (__any(pend))
                        if (pend && ccm::atan2f_exact(cfg.max_distance, w8[u]) < tab_u)
                            oo[row] |= 0x80;
                }
            }
        }
    }
    __syncthreads();
    if (!(g.debug_flags & 4))
    {
        int lc = lc0;
        for (int c = 0; c < ncols; c++)
        {
            for (int row = lane; row < R; row += 64)
            {
                const size_t ci = (size_t) lc * R + row;
                const unsigned char o = l_out[c * PB + row];
                // label codes -> the reference's label values by shifts of packed constants (a table in memory would cost two more
                // loads per cell)
                constexpr unsigned long long GV = (unsigned long long) CC_GP_UNKNOWN | ((unsigned long long) CC_GP_GROUND << 8) |
                                                  ((unsigned long long) CC_GP_OBSTACLE << 16) | ((unsigned long long) CC_GP_EGO_VEHICLE << 24) |
                                                  ((unsigned long long) CC_GP_FOG << 32);
                constexpr unsigned long long DV0 = (unsigned long long) CC_DBG_WHITE | ((unsigned long long) CC_DBG_GRAY << 8) |
                                                   ((unsigned long long) CC_DBG_ORANGE << 16) | ((unsigned long long) CC_DBG_GREEN << 24) |
                                                   ((unsigned long long) CC_DBG_YELLOWGREEN << 32) | ((unsigned long long) CC_DBG_YELLOW << 40) |
                                                   ((unsigned long long) CC_DBG_RED << 48) | ((unsigned long long) CC_DBG_DARKRED << 56);
                constexpr unsigned DV1 = (unsigned) CC_DBG_VIOLET | ((unsigned) CC_DBG_LIGHTGRAY << 8);
                const unsigned dcode = (o >> 3) & 15;
                p.ground[ci] = (unsigned char) (GV >> (8 * (o & 7)));
                p.debug[ci] = (unsigned char) (dcode < 8 ? (DV0 >> (8 * dcode)) : (unsigned long long) (DV1 >> (8 * (dcode - 8))));
                // cc.cpp:567-616: everything that is not an obstacle is ignored, and so are the filtered obstacles
                const bool ign = (o & 7) != SG_G_OBSTACLE || (o & 0x80);
                p.ignored[ci] = ign ? 1 : 0;
            }
            lc = lc + 1 == RC ? 0 : lc + 1;
        }
    }
}

// =====================================================================================================
// k_seg_small — the whole ground segmentation of a column (cc.cpp:294-624) by ONE wavefront with lanes = ROWS, for calls of a few firings
// (the per-column latency path: BASELINE.json configs[1]). Round 4.
//
// k_seg_scan walks a column bottom-up on one lane — fine when 64 columns share the wavefront, 20 - 40 us when a call brings one column: 64 rows x
// ~250 dependent instructions on a lone wavefront. Here the rows are the lanes and the row-serial state machine is solved as a FIXED POINT:
//   * what does not depend on the labels below is computed once, for all rows at once: the previous point outside the ego box (nearest valid row
//     below: one ballot + find-first-set + lane permute), the slope to it, "flat w.r.t. previous", the first point's test, the geometric part of
//     the last-ground-point rule (cc.cpp:546-548);
//   * the state a row sees — first_obstacle_detected, last_ground_position, previous_label — is a function of the LABELS of the rows below it:
//     "some row below is RED (or the first point was an obstacle)", "the nearest row below that updates the last ground point", "the label of
//     the previous valid row". Given a guess of all labels, every row recomputes its own label from the guess; rows only depend on rows below, so
//     after k rounds the lowest k valid rows are final and the iteration ends at the unique sequential solution, in at most `rows` rounds — on
//     real columns after 3 - 6 (ground, then one or two obstacle / ground changes);
//   * the downward fix-up of cc.cpp:513-535 (ground cells right below a new obstacle become obstacles) only reaches down to the next RED row, so
//     the walks of different RED rows are disjoint: a cell is converted iff every cell between it and the nearest RED row above passes the
//     walk's test — one ballot and two mask operations.
// The table of inclination steps needs no tiles here: the stream's table as of the previous column is Planes::curtab (rows = lanes).
// One wavefront per stream, the batch's columns in order (a call of n firings finishes about n columns). Reads the cells from the ring like
// k_seg_pre; writes labels, ignore flags, tags, the records / inclinations of cells without a return, column entries, curtab. No staging planes.
// grid = streams, block = 64; num_rows <= 64.
// =====================================================================================================
__device__ __forceinline__ void seg_small_body(const Geometry& g, const cc_config& cfg, const Planes& P, StreamState* states, int first_stream, int slot,
                                               const double* __restrict__ poses, long long n_total, long long fbase, const double* __restrict__ ego,
                                               long long n_batch, const int sl)
{
    const int s = first_stream + sl;
    StreamState* st = &states[s];
    const int lane = lane_id();
    if (lane == 0)
        st->batch[slot].mode = st->assoc_mode; // (what k_table does first)
    const long long seg_begin = st->batch[slot].seg_begin, seg_end = st->batch[slot].seg_end;
    if (seg_begin < 0 || seg_begin >= seg_end || st->error != 0)
        return;
    if (!st->has_robot_tf)
    {
        if (lane == 0)
            raise_error(st, CC_ERR_NO_ROBOT_TRANSFORM, seg_begin, 0);
        return;
    }
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols, NC = g.num_columns;
    const int row = lane;
    const bool inrow = row < R;
    const float height_sensor_to_ground = -(float) st->robot_from_sensor[11] + cfg.height_ref_to_ground_;
    const bool supplement = cfg.supplement_inclination_angle_for_nan_cells != 0;
    const bool step_filter = cfg.ignore_points_with_too_big_inclination_angle_diff != 0;
    const bool no_terrain = cfg.use_terrain == 0;
    // lane masks: rows strictly below this one (= larger row index, visited earlier by the bottom-up walk) / strictly above
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const unsigned long long below = ~le, above = le >> 1;
    float tabrow = inrow ? p.curtab[row] : 0.f; // sc_inclination_angles_between_lasers_[row] as of the previous column
    int lc = (int) (seg_begin % RC);
    long long rot = seg_begin / NC;
    int cir = (int) (seg_begin - rot * NC);
    long long pass = seg_begin / RC;
    CazBase cb = caz_base_of_rotation(rot);
    for (long long gc = seg_begin; gc < seg_end; gc++)
    {
        const size_t ci = (size_t) lc * R + row;
        const uint16_t tag = cell_tag(pass);
        // ---- the column's cells (as k_seg_pre reads them)
        float cx[1] = {0.f}, cy[1] = {0.f}, cz[1] = {0.f}, dist[1] = {__builtin_nanf("")}, incl[1] = {__builtin_nanf("")};
        uint8_t inten[1] = {0};
        float incaz = 0.f;
        bool empty_cell = false, overrun = false;
        long long overrun_gcol = -1;
        if (inrow)
        {
            const uint16_t tg = p.gtag[ci];
            dist[0] = p.dist[ci];
            if (tg == tag)
            {
                const float4 r4 = p.sc_rec[ci];
                cx[0] = r4.x, cy[0] = r4.y, cz[0] = r4.z, incl[0] = r4.w;
                incaz = p.incaz[ci];
                inten[0] = p.inten[ci];
            }
            else
            {
                empty_cell = true;
                if (tg != CELL_CLEARED)
                {
                    overrun = true; // cc.cpp:320-345
                    overrun_gcol = gc - (long long) ((((unsigned) tag - (unsigned) tg) & 0x7fffu)) * RC;
                }
            }
        }
        if (__any(overrun))
        {
            const int worst = -wave_min_i32(overrun ? -row : 1); // the highest stale row = the first one of the reference's bottom-up walk
            if (overrun && row == worst)
            {
                atomicMin((unsigned long long*) &st->overrun_col, (unsigned long long) gc);
                raise_error(st, CC_ERR_RING_OVERRUN, overrun_gcol, gc);
            }
            break; // (the reference throws here: nothing behind this column is segmented; the host reports the error)
        }
        const int trig = uniform_i32(p.trig[lc]);
        const double* T = poses + ((size_t) sl * (size_t) n_total + (size_t) fbase + (size_t) trig) * 12;
        const double* E = ego + ((size_t) sl * (size_t) n_batch + (size_t) trig) * EGO_STRIDE;
        float x2a[1], uza[1], wa[1];
        int fla[1];
        seg_pre_cells<1>(cfg, R, lane, cx, cy, cz, dist, incl, inten, (float) T[3], (float) T[7], (float) T[11], E, x2a, uza, wa, fla);
        const int f = fla[0];
        const float cur2x = x2a[0], cur2y = uza[0], wv = wa[0];
        // ---- the table as of this column, supplemented inclinations (cc.cpp:353-369), pending inclination-step tests (:597-603)
        const bool is_nan = (f & SG_NAN) != 0;
        const bool own = !(f & (SG_NAN | SG_PENDING));
        const float tab = own ? wv : tabrow;
        tabrow = tab;
        float sincl = incl[0];                                        // inclination the cell ends up with
        bool done = !is_nan | !supplement | (row >= R - 1) | !inrow; // (a cell without a return in the last row keeps NaN)
        while (__any(!done))
        {
            // runs of cells without a return resolve bottom-up, one row per round: the row below first (its value AFTER the supplement)
            const float sb = __shfl_down(sincl, 1, 64);
            const int db = __shfl_down(done ? 1 : 0, 1, 64);
            if (!done && db)
            {
                sincl = sb + tab;
                done = true;
            }
        }
        bool ign = (f & (SG_TOO_CLOSE | SG_INCL_IGNORE)) != 0;
        {
            const bool pend = ((f & (SG_PENDING | SG_NAN)) == SG_PENDING) & step_filter & (row < R - 1) & !(tab != tab);
            const float a = wv * tab; // (wv: the distance of a pending cell)
            const bool in_range = (cfg.max_distance > 0.f) & (tab >= 0.f) & (tab < 0.05f) & (wv > 0.f) & (wv < 3.0e38f);
            const bool surely_not = in_range & (cfg.max_distance >= 1.01f * a);
            const bool surely = in_range & !surely_not & (cfg.max_distance * 1.000002f < a);
            const bool need = pend & !surely_not & !surely;
            ign |= pend & surely;
            if (__any(need))
                if (need && ccm::atan2f_exact(cfg.max_distance, wv) < tab)
                    ign = true;
        }
        if (cfg.ignore_points_in_chessboard_pattern)
            ign |= ((gc & 1) != 0) != ((row & 1) != 0); // cc.cpp:600-606: column parity differs from row parity
        // ---- state machine, label-independent part
        const bool valid = inrow & ((f & (SG_NAN | SG_FOG | SG_EGO)) == 0);
        const unsigned long long V = __ballot(valid);
        const unsigned long long mb = V & below;
        const bool has_prev = mb != 0;
        const int pb = has_prev ? __ffsll((long long) mb) - 1 : lane; // previous point outside the ego box = nearest valid row below
        const bool first = valid & !has_prev, normal = valid & has_prev;
        const float pv2x = __shfl(cur2x, pb, 64), pvz = __shfl(cur2y, pb, 64);
        const float p2cx = cur2x - pv2x, p2cy = cur2y - pvz;
        const float slope_to_prev = p2cy / p2cx;
        const bool flat_prev = (ccm::absf(slope_to_prev) < cfg.max_slope) & (p2cx > 0) & (no_terrain | (p2cx < 5));
        const bool keep_geo = (slope_to_prev > cfg.last_ground_point_slope_higher_than) & (ccm::absf(p2cx) < cfg.last_ground_point_distance_smaller_than);
        const float h = cur2y - height_sensor_to_ground;
        const bool first_ground = (h > cfg.first_ring_as_ground_min_allowed_z_diff) & (h < cfg.first_ring_as_ground_max_allowed_z_diff);
        const bool first_obst = __any(first & !first_ground);
        // ---- fixed point over the labels (debug codes; the ground label follows from them)
        unsigned d = first ? (first_ground ? (unsigned) SG_D_GRAY : (unsigned) SG_D_ORANGE)
                           : (normal ? (flat_prev ? (unsigned) SG_D_GREEN : (unsigned) SG_D_RED) : (unsigned) SG_D_WHITE);
        for (int round = 0; round <= R; round++)
        {
            const unsigned long long REDm = __ballot(normal & (d == (unsigned) SG_D_RED));
            const unsigned long long YELm = __ballot(normal & (d == (unsigned) SG_D_YELLOW));
            const bool fod = first_obst | ((REDm & below) != 0);                      // first_obstacle_detected as this row sees it
            const bool prev_yellow = has_prev & (((YELm >> pb) & 1ull) != 0);         // previous_label == YELLOW
            const bool upd = (first & first_ground) | (normal & ((d == (unsigned) SG_D_GREEN) | (d == (unsigned) SG_D_YELLOWGREEN)) & keep_geo & !prev_yellow);
            const unsigned long long ml = __ballot(upd) & below;
            const bool has_lg = ml != 0;
            const int lgrow = has_lg ? __ffsll((long long) ml) - 1 : lane;            // the row that set last_ground_position
            const float lgx = __shfl(cur2x, lgrow, 64), lgy = __shfl(cur2y, lgrow, 64);
            const float lg2x = has_lg ? lgx : 0.f, lgz = has_lg ? lgy : height_sensor_to_ground;
            const float l2cx = cur2x - lg2x, l2cy = cur2y - lgz;
            const float slope_to_lg = l2cy / l2cx;
            const bool flat_lg = (ccm::absf(slope_to_lg) < cfg.max_slope) & (l2cx > 0);
            const bool green = !fod & flat_prev;
            const bool yellowgreen = !green & no_terrain & fod & flat_prev & flat_lg;
            const bool yellow = !green & !yellowgreen & no_terrain & (ccm::absf(l2cx) < cfg.ground_because_close_to_last_certain_ground_max_dist_diff) &
                                (ccm::absf(l2cy) < cfg.ground_because_close_to_last_certain_ground_max_z_diff);
            const unsigned dn = normal ? (green ? (unsigned) SG_D_GREEN
                                                : (yellowgreen ? (unsigned) SG_D_YELLOWGREEN : (yellow ? (unsigned) SG_D_YELLOW : (unsigned) SG_D_RED)))
                                       : d;
            const bool changed = dn != d;
            d = dn;
            if (!__any(changed))
                break;
        }
        unsigned gl = (d == (unsigned) SG_D_ORANGE || d == (unsigned) SG_D_RED) ? (unsigned) SG_G_OBSTACLE : (unsigned) SG_G_GROUND;
        gl = valid ? gl : (unsigned) SG_G_UNKNOWN;
        if (f & SG_EGO)
        {
            gl = SG_G_EGO;
            d = SG_D_VIOLET;
        }
        if (f & SG_FOG)
        {
            gl = SG_G_FOG;
            d = SG_D_LIGHTGRAY;
        }
        if ((f & SG_NAN) || !inrow)
        {
            gl = SG_G_UNKNOWN;
            d = SG_D_WHITE;
        }
        // ---- downward fix-up (cc.cpp:513-535): ground cells (and YELLOW ones) right below a RED row, as far as every cell passes the test
        {
            const unsigned long long REDm = __ballot(normal & (d == (unsigned) SG_D_RED));
            const unsigned long long RA = REDm & above; // RED rows above this cell
            const bool has_ra = RA != 0;
            const int ra = has_ra ? 63 - __clzll((long long) RA) : lane; // the nearest one: its walk is the only one that can get here
            const float xr = __shfl(cur2x, ra, 64);
            const bool pass_test = has_ra & ((d == (unsigned) SG_D_YELLOW) |
                                             ((gl == (unsigned) SG_G_GROUND) & (ccm::absf(xr - cur2x) < cfg.obstacle_because_next_certain_obstacle_max_dist_diff)));
            const unsigned long long NP = __ballot(!pass_test);
            const unsigned long long le_ra = ra == 63 ? ~0ull : ((2ull << ra) - 1ull);
            const bool reached = has_ra & ((NP & le & ~le_ra) == 0); // every cell of (ra, this row] passes
            if (reached & (gl == (unsigned) SG_G_GROUND))
            {
                gl = SG_G_OBSTACLE;
                d = SG_D_DARKRED;
            }
        }
        // ---- results
        if (inrow)
        {
            constexpr unsigned long long GV = (unsigned long long) CC_GP_UNKNOWN | ((unsigned long long) CC_GP_GROUND << 8) |
                                              ((unsigned long long) CC_GP_OBSTACLE << 16) | ((unsigned long long) CC_GP_EGO_VEHICLE << 24) |
                                              ((unsigned long long) CC_GP_FOG << 32);
            constexpr unsigned long long DV0 = (unsigned long long) CC_DBG_WHITE | ((unsigned long long) CC_DBG_GRAY << 8) |
                                               ((unsigned long long) CC_DBG_ORANGE << 16) | ((unsigned long long) CC_DBG_GREEN << 24) |
                                               ((unsigned long long) CC_DBG_YELLOWGREEN << 32) | ((unsigned long long) CC_DBG_YELLOW << 40) |
                                               ((unsigned long long) CC_DBG_RED << 48) | ((unsigned long long) CC_DBG_DARKRED << 56);
            constexpr unsigned DV1 = (unsigned) CC_DBG_VIOLET | ((unsigned) CC_DBG_LIGHTGRAY << 8);
            p.ground[ci] = (unsigned char) (GV >> (8 * gl));
            p.debug[ci] = (unsigned char) (d < 8 ? (DV0 >> (8 * d)) : (unsigned long long) (DV1 >> (8 * (d - 8))));
            p.ignored[ci] = (gl != (unsigned) SG_G_OBSTACLE || ign) ? 1 : 0; // cc.cpp:567-616
            if (empty_cell)
                p.gtag[ci] = tag;
            if (is_nan)
            {
                p.sc_rec[ci] = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), sincl);
                p.incl[ci] = sincl;
            }
        }
        int kpos = 0x7fffffff, kneg = 0x7fffffff;
        if (inrow && !is_nan)
            caz_key(incaz, kpos, kneg);
        const double min_az = column_min_caz(cb, kpos, kneg, inrow && is_nan, gc, g.az_width);
        if (lane == 0)
        {
            p.colg[lc] = gc;
            p.colminaz[lc] = min_az;
        }
        // next column
        lc = lc + 1 == RC ? 0 : lc + 1;
        pass += lc == 0 ? 1 : 0;
        cir = cir + 1 == NC ? 0 : cir + 1;
        if (cir == 0)
        {
            rot++;
            cb = caz_base_of_rotation(rot);
        }
    }
    if (inrow)
        p.curtab[row] = tabrow;
}

__global__ __launch_bounds__(64) void k_seg_small(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot,
                                                  const double* __restrict__ poses, long long n_total, long long fbase, const double* __restrict__ ego,
                                                  long long n_batch)
{
    seg_small_body(g, cfg, P, states, first_stream, slot, poses, n_total, fbase, ego, n_batch, (int) blockIdx.x);
}

// =====================================================================================================
// k_associate — continuous_clustering.cpp:638-1145. One wavefront per stream, lanes = rows.
// =====================================================================================================
constexpr int LINK_SLOTS_V1 = 8;

struct AssocCtx
{
    SP p;
    int R, NC, RC;
    float az_width, maxd2;
    int max_steps_in_row, max_steps_in_column, stop_enabled, stop_min_steps;
};

// lock-free union-find over tree roots (cell indices); every access bypasses L1
__device__ __forceinline__ int uf_find(int32_t* uf, int a)
{
    while (true)
    {
        const int pa = ld_agent(&uf[a]);
        if (pa == a)
            return a;
        const int gp = ld_agent(&uf[pa]);
        if (gp != pa)
            __hip_atomic_store(&uf[a], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // path halving
        a = pa;
    }
}

__device__ __forceinline__ void uf_union(int32_t* uf, int a, int b)
{
    while (true)
    {
        a = uf_find(uf, a);
        b = uf_find(uf, b);
        if (a == b)
            return;
        if (a < b)
        {
            const int t = a;
            a = b;
            b = t;
        }
        // hang the larger index under the smaller one
        if (atomicCAS(&uf[a], a, b) == a)
            return;
    }
}

__device__ __forceinline__ void tree_init(const SP& p, int cell, double fin)
{
    p.t_fin[cell] = fin;
    p.t_width[cell] = 1;
    p.t_pts[cell] = 1;
    p.t_uf[cell] = cell;
    p.t_cid[cell] = 0;
    p.t_finished[cell] = 0;
}

// The window scan of traverseFieldOfView (cc.cpp:698-771) for one point.
//  LIVE = false: record the first passing candidate as `parent` and later passing candidates as link candidates;
//                no tree state is read (valid when no attach is refused, checked by the caller).
//  LIVE = true : exact reference semantics with immediate attach / link (single lane, rows in order).
struct NoLinkVisitor
{
    __device__ __forceinline__ void operator()(int) const {}
};
// (ON_LINK, static scan only: called with every accepted candidate behind the first, in the reference's order, whether or not it still fits `links`:
// k_assocb walks the complete list of a point whose recorded list overflowed with it)
template<bool LIVE, bool CODE = false, bool REC = false, class ON_LINK = NoLinkVisitor>
__device__ __forceinline__ void scan_point(const AssocCtx& c, const int lc, const long long gc, const int row, const int first_local,
                                           const float mad, const double pcaz, int& p_root, int& parent, int* links, int& nlinks,
                                           bool& overflow, const int max_links = LINK_SLOTS_V1, int* visits = nullptr, StreamState* st = nullptr,
                                           const Geometry* geo = nullptr, int* reach = nullptr, const ON_LINK& on_link = ON_LINK())
{
    const SP& p = c.p;
    const int R = c.R;
    const int pi = lc * R + row;
    const float4 me = p.sc_rec[pi]; // (REC only says how the visited cells are read: the records are the one copy of x, y, z)
    const float pincl = me.w, px = me.x, py = me.y, pz = me.z;
    int needed = f2i_x86(__builtin_ceilf(mad / c.az_width));
    needed = needed < c.max_steps_in_row ? needed : c.max_steps_in_row;
    int oc = lc;
    bool rooted = LIVE ? (p_root != -1) : false;
    for (int sb = 0; sb <= needed; sb++)
    {
        for (int dir = -1; dir <= 1; dir += 2)
        {
            if (dir == 1 && sb == 0)
                continue;
            int sv = (dir == 1 || sb == 0) ? 1 : 0;
            int orow = (dir == 1 || sb == 0) ? row + dir : row;
            while (orow >= 0 && orow < R && sv <= c.max_steps_in_column)
            {
                const int oi = oc * R + orow;
                if (visits)
                    ++*visits; // cc.cpp:725
                if (reach)
                    *reach = sb;
                const float4 orec = p.sc_rec[oi];
                unsigned char oign = 0;
                if (REC)
                    oign = p.ignored[oi]; // both loads are issued before the first use: one round trip per visit
                const float oincl = orec.w;
                if (ccm::absf(oincl - pincl) > mad)
                    break;
                if (REC ? !oign : !p.ignored[oi])
                {
                    bool consider = true;
                    int oroot = -1;
                    if (LIVE)
                    {
                        oroot = p.root[oi];
                        consider = (p_root >= 0 && p_root / R == 0) || oroot != p_root; // cc.cpp:733 incl. its "== 0" quirk
                    }
                    if (consider)
                    {
                        const float dx = px - orec.x, dy = py - orec.y, dz = pz - orec.z;
                        if (dx * dx + dy * dy + dz * dz < c.maxd2)
                        {
                            if (LIVE)
                            {
                                if (p_root == -1)
                                {
                                    // associatePointToPointTree cc.cpp:643-673
                                    const long long rg = p.colg[oroot / R];
                                    const uint32_t nw = (uint32_t) (gc - rg + 1);
                                    if (nw <= (uint32_t) c.NC && !p.t_finished[oroot])
                                    {
                                        p_root = oroot;
                                        parent = (sb << 8) | orow; // the point joins other's child list (cc.cpp:663)
                                        p.t_width[oroot] = nw;
                                        const double cand = pcaz + (double) mad;
                                        const double cur = ld_agent(&p.t_fin[oroot]);
                                        if (cand > cur)
                                            __hip_atomic_store(&p.t_fin[oroot], cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        atomicAdd(&p.t_pts[oroot], 1u);
                                    }
                                }
                                else
                                {
                                    // associatePointTreeToPointTree cc.cpp:675-696
                                    if (!p.t_finished[p_root] && !p.t_finished[oroot] && p_root != oroot)
                                    {
                                        if (geo)
                                            log_link(*geo, st, p.link_log, p_root, oroot);
                                        uf_union(p.t_uf, p_root, oroot);
                                    }
                                }
                            }
                            else
                            {
                                const int cand = CODE ? ((sb << 8) | orow) : oi;
                                if (!rooted)
                                {
                                    parent = cand;
                                    rooted = true;
                                }
                                else
                                {
                                    on_link(cand);
                                    if (nlinks < max_links)
                                        links[nlinks++] = cand;
                                    else
                                        overflow = true;
                                }
                            }
                        }
                    }
                }
                if (LIVE)
                    rooted = p_root != -1;
                if (rooted && c.stop_enabled && sv >= c.stop_min_steps)
                    break;
                orow += dir;
                sv++;
            }
        }
        if (rooted && c.stop_enabled && sb >= c.stop_min_steps)
            break;
        if (oc == first_local)
            break;
        oc--;
        if (oc < 0)
            oc += c.RC;
    }
}

// what __syncthreads() is for a block of one wavefront, without the barrier instruction: every earlier global / LDS access of the wavefront has completed
// before a later one is issued (lanes hand values to each other through memory between the phases of k_associate)
__device__ __forceinline__ void assoc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One stream's batch in global memory, one wavefront (lanes = rows). Called by k_associate (a block = one wavefront = one stream) and, behind the serial
// LDS kernel, by wavefront 0 of k_assoc3's block (cc_assoc3.h): no block barrier in here — assoc_wave_sync() orders the wavefront's own global and LDS
// accesses the way __syncthreads() does for a one-wavefront block.
template<int RPL>
__device__ __forceinline__ void associate_stream(const Geometry& g, const cc_config& cfg, const Planes& P, StreamState* states, const int s, const int slot)
{
    const int lane = lane_id();
    StreamState* st = &states[s];
    if (st->error != 0 || st->batch[slot].seg_begin < 0 || (st->assoc_mode == 0 && st->batch[slot].mode == 0) ||
        st->batch[slot].acp_next >= st->batch[slot].seg_end)
        return; // (the LDS kernels take batches that were staged for them, unless the stream overflowed their tree pool meanwhile)
    AssocCtx c;
    c.p = stream_ptrs(P, g, s);
    const SP& p = c.p;
    const int R = c.R = g.num_rows;
    const int NC = c.NC = g.num_columns;
    const int RC = c.RC = g.ring_cols;
    c.az_width = g.az_width;
    c.maxd2 = g.max_distance_squared;
    c.max_steps_in_row = cfg.max_steps_in_row;
    c.max_steps_in_column = cfg.max_steps_in_column;
    c.stop_enabled = cfg.stop_after_association_enabled;
    c.stop_min_steps = cfg.stop_after_association_min_steps;
    const int nth = cfg.cluster_point_trees_every_nth_column;

    __shared__ int s_parent[WAVE * MAX_ROWS_PER_LANE];
    __shared__ int s_links[WAVE * MAX_ROWS_PER_LANE][LINK_SLOTS_V1];
    __shared__ int s_bcast[4];
    __shared__ double s_bd[2];
    __shared__ long long s_bl[2];

    long long first_unpub = st->first_unpublished, ring_start = st->ring_start;
    if (lane == 0 && st->batch[slot].pub_begin < 0)
        st->batch[slot].pub_begin = first_unpub; // first association kernel of this pass
    unsigned long long cluster_counter = st->cluster_counter;
    int n_unf = st->n_unfinished;
    long long M = st->min_required;
    double L = st->finish_lower_bound;
    double last_min_az = st->last_round_min_az;
    unsigned long long cells_published = st->cells_published, clusters_finished = st->clusters_finished;
    unsigned long long exceed = st->exceed_one_rotation, serial_cols = st->serial_columns, alias_rounds = st->stamp_alias_rounds;
    int n_events = st->n_events;
    const long long col_end = st->batch[slot].seg_end;
    int err = 0;
    long long err_a = 0, err_b = 0;

    auto emit = [&](int type, long long a, long long b, unsigned cc, unsigned dd, long long column)
    {
        if (!g.record_events)
            return;
        if (lane == 0)
        {
            if (n_events < g.event_capacity)
            {
                cc_event e;
                e.type = type;
                e.stream = s;
                e.a = a;
                e.b = b;
                e.c = cc;
                e.d = dd;
                e.column = column;
                p.events[n_events] = e;
            }
        }
        n_events++;
    };

    for (long long gc = st->batch[slot].acp_next; gc < col_end && err == 0; gc++)
    {
        const int lc = (int) (gc % RC);
        const int first_local = (int) (first_unpub % RC);
        const CazBase cb = caz_base_of_column(gc, g.num_columns);
        emit(CC_EV_GROUND_COLUMN, gc, gc, 0, 0, gc);

        // ------------------------------------------------------------------ association (cc.cpp:773-835)
        float mad[RPL];
        double pcaz[RPL];
        bool active[RPL];
        int parent[RPL], nlinks[RPL];
        bool overflow = false;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            active[k] = false;
            parent[k] = -1;
            nlinks[k] = 0;
            mad[k] = 0.f;
            pcaz[k] = 0.;
            if (row < R)
            {
                const int ci = lc * R + row;
                if (!p.ignored[ci])
                {
                    active[k] = true;
                    mad[k] = ccm::asinf_exact(cfg.max_distance / p.dist[ci]);
                    pcaz[k] = cell_caz(cb, p.incaz[ci]);
                    int dummy_root = -1, vis = 0;
                    scan_point<false>(c, lc, gc, row, first_local, mad[k], pcaz[k], dummy_root, parent[k], s_links[row], nlinks[k],
                                      overflow, LINK_SLOTS_V1, &vis);
                    if (g.mirror_fields)
                        p.sc_visits[ci] = sat_u16(vis);
                }
                s_parent[row] = active[k] ? parent[k] : -2;
                // this kernel takes its candidates as cell indices; the planes keep the (columns back, row) code of k_scan
                {
                    int code = active[k] ? -1 : -2;
                    if (parent[k] >= 0)
                    {
                        int back = lc - parent[k] / R;
                        back = back < 0 ? back + RC : back;
                        code = (back << 8) | (parent[k] % R);
                    }
                    p.sc_parent[ci] = (int16_t) code;
                    if (!active[k] && g.mirror_fields)
                        p.sc_visits[ci] = 0;
                }
            }
        }
        assoc_wave_sync();
        // resolve tree roots through same-column parents, then verify that no attach would have been refused
        int rootc[RPL];
        bool refused = overflow;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            rootc[k] = -1;
            if (active[k])
            {
                if (parent[k] < 0)
                    rootc[k] = lc * R + row; // new tree
                else
                {
                    int r = parent[k];
                    while (true)
                    {
                        if (r / R != lc)
                        {
                            r = p.root[r];
                            break;
                        }
                        const int pr = s_parent[r - lc * R];
                        if (pr < 0)
                            break; // r is a new tree root of this column
                        r = pr;
                    }
                    rootc[k] = r;
                    if (r < 0)
                        refused = true; // candidate without tree: impossible for a processed, non-ignored cell
                    else
                    {
                        const long long rg = p.colg[r / R];
                        const uint32_t nw = (uint32_t) (gc - rg + 1);
                        if (nw > (uint32_t) NC || p.t_finished[r])
                            refused = true;
                    }
                }
            }
        }
        const bool column_serial = __any(refused);

        if (!column_serial)
        {
            // (a) roots + new trees in row order
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                const bool is_new = active[k] && parent[k] < 0;
                const unsigned long long mask = __ballot(is_new);
                const int cnt = __popcll(mask);
                if (n_unf + cnt > g.tree_capacity)
                {
                    err = CC_ERR_CAPACITY;
                    err_a = n_unf + cnt;
                    break;
                }
                if (row < R)
                    p.root[lc * R + row] = active[k] ? rootc[k] : -1;
                if (is_new)
                {
                    const int cell = lc * R + row;
                    const int pos = n_unf + __popcll(mask & lanes_below());
                    const double fin = pcaz[k] + (double) mad[k];
                    tree_init(p, cell, fin);
                    p.ulist[pos] = cell;
                    p.t_pos[cell] = pos;
                    L = fin < L ? fin : L;
                }
                if (cnt > 0 && n_unf == 0)
                    M = gc;
                n_unf += cnt;
            }
            L = wave_min_f64(L);
            assoc_wave_sync();
            // (b) attach: root bookkeeping of associatePointToPointTree (cc.cpp:661-671)
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                if (active[k] && parent[k] >= 0)
                {
                    const int r = rootc[k];
                    const long long rg = p.colg[r / R];
                    p.t_width[r] = (uint32_t) (gc - rg + 1);
                    const double cand = pcaz[k] + (double) mad[k];
                    atomicMax((unsigned long long*) &p.t_fin[r], (unsigned long long) __double_as_longlong(cand));
                    atomicAdd(&p.t_pts[r], 1u);
                }
            }
            assoc_wave_sync();
            // (c) links between trees (cc.cpp:675-696) as lock-free unions
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (active[k] && parent[k] >= 0)
                {
                    const int rp = rootc[k];
                    for (int j = 0; j < nlinks[k]; j++)
                    {
                        const int rq = p.root[s_links[row][j]];
                        if (rq != rp && rq >= 0 && !p.t_finished[rp] && !p.t_finished[rq])
                        {
                            log_link(g, st, p.link_log, rp, rq);
                            uf_union(p.t_uf, rp, rq);
                        }
                    }
                }
            }
            assoc_wave_sync();
        }
        else
        {
            // exact serial replay of the column by one lane (rare: >1-rotation clusters, finished trees in reach)
            serial_cols++;
            if (lane == 0)
            {
                int nn = n_unf;
                double LL = L;
                long long MM = M;
                int e = 0;
                for (int row = 0; row < R; row++)
                {
                    const int ci = lc * R + row;
                    if (p.ignored[ci])
                    {
                        p.root[ci] = -1;
                        p.sc_parent[ci] = -2;
                        if (g.mirror_fields)
                            p.sc_visits[ci] = 0;
                        continue;
                    }
                    const float m = ccm::asinf_exact(cfg.max_distance / p.dist[ci]);
                    const double caz = cell_caz(cb, p.incaz[ci]);
                    int proot = -1, par = -1, nl = 0, vis = 0;
                    bool ov = false;
                    scan_point<true>(c, lc, gc, row, first_local, m, caz, proot, par, nullptr, nl, ov, LINK_SLOTS_V1, &vis, st, &g);
                    p.sc_parent[ci] = (int16_t) par; // the live scan's parent replaces the static one
                    if (g.mirror_fields)
                        p.sc_visits[ci] = sat_u16(vis);
                    if (proot == -1)
                    {
                        if (nn + 1 > g.tree_capacity)
                        {
                            e = CC_ERR_CAPACITY;
                            break;
                        }
                        proot = ci;
                        const double fin = caz + (double) m;
                        tree_init(p, ci, fin);
                        p.ulist[nn] = ci;
                        p.t_pos[ci] = nn;
                        if (nn == 0)
                            MM = gc;
                        nn++;
                        LL = fin < LL ? fin : LL;
                    }
                    p.root[ci] = proot;
                }
                s_bcast[0] = nn;
                s_bcast[1] = e;
                s_bd[0] = LL;
                s_bl[0] = MM;
            }
            assoc_wave_sync();
            n_unf = s_bcast[0];
            if (s_bcast[1])
            {
                err = s_bcast[1];
                err_a = n_unf;
            }
            L = s_bd[0];
            M = s_bl[0];
            assoc_wave_sync();
        }
        if (err)
            break;

        // ------------------------------------------------------------------ finished-cluster check (cc.cpp:837-974)
        if (gc % nth != 0)
            continue;
        const double min_az = p.colminaz[lc];
        long long M_c;
        if (n_unf == 0)
            M_c = gc + 1;
        else if (min_az == last_min_az)
        {
            // every older tree still carries the visited stamp of the previous round: none of them
            // is a BFS start and none is expanded; trees created in this column cannot be finished yet.
            alias_rounds++;
            M_c = M;
        }
        else if (!(min_az >= L) && !((gc + 1 - M) >= NC))
            M_c = M; // no cluster can be finished: nothing to scan
        else
        {
            // full pass over the unfinished trees
            for (int i = lane; i < n_unf; i += 64)
            {
                p.agg_fin[i] = 0ull;
                p.agg_min[i] = 0x7fffffffffffffffll;
                p.agg_max[i] = 0;
                p.agg_pts[i] = 0;
                p.agg_first[i] = 0x7fffffff;
                p.agg_cid[i] = 0;
                p.agg_flag[i] = 0;
            }
            assoc_wave_sync();
            for (int i = lane; i < n_unf; i += 64)
            {
                const int t = p.ulist[i];
                const int rep = uf_find(p.t_uf, t);
                const int j = p.t_pos[rep];
                p.ucomp[i] = j;
                const long long tg = p.colg[t / R];
                atomicMax(&p.agg_fin[j], (unsigned long long) __double_as_longlong(ld_agent(&p.t_fin[t])));
                atomicMin(&p.agg_min[j], tg);
                atomicMax(&p.agg_max[j], tg + (long long) p.t_width[t]);
                atomicAdd(&p.agg_pts[j], ld_agent(&p.t_pts[t]));
                atomicMin(&p.agg_first[j], i);
            }
            assoc_wave_sync();
            int exceed_local = 0;
            for (int i = lane; i < n_unf; i += 64)
            {
                if (p.ucomp[i] == i)
                {
                    const double fin = __longlong_as_double((long long) ld_agent(&p.agg_fin[i]));
                    const bool unfinished = fin > min_az;
                    const bool exceeds = (ld_agent(&p.agg_max[i]) - ld_agent(&p.agg_min[i])) >= NC;
                    if (exceeds)
                        exceed_local++;
                    p.agg_flag[i] = (!unfinished || exceeds) ? 1 : 0;
                }
            }
            for (int o = 32; o > 0; o >>= 1)
                exceed_local += __shfl_xor(exceed_local, o);
            exceed += exceed_local;
            assoc_wave_sync();
            // ids in the order the reference's BFS would discover the clusters: by earliest tree in the list
            int last_first = -1;
            while (true)
            {
                int best = 0x7fffffff;
                for (int i = lane; i < n_unf; i += 64)
                    if (p.ucomp[i] == i && p.agg_flag[i] && ld_agent(&p.agg_pts[i]) > 5u)
                    {
                        const int fi = ld_agent(&p.agg_first[i]);
                        if (fi > last_first && fi < best)
                            best = fi;
                    }
                best = wave_min_i32(best);
                if (best == 0x7fffffff)
                    break;
                // the representative's slot is ucomp[best]
                const int j = p.ucomp[best];
                const unsigned cid = (unsigned) cluster_counter;
                if (lane == 0)
                    p.agg_cid[j] = cid;
                emit(CC_EV_CLUSTER, ld_agent(&p.agg_min[j]), ld_agent(&p.agg_max[j]) - 1, cid, ld_agent(&p.agg_pts[j]), gc);
                cluster_counter++;
                clusters_finished++;
                last_first = best;
            }
            assoc_wave_sync();
            // mark trees, minimum required column, stable compaction of the list
            long long min_all = 0x7fffffffffffffffll, min_surv = 0x7fffffffffffffffll;
            double L_new = 1.7976931348623157e308;
            int out = 0;
            for (int base = 0; base < n_unf; base += 64)
            {
                const int i = base + lane;
                bool surv = false;
                int t = -1;
                if (i < n_unf)
                {
                    t = p.ulist[i];
                    const int j = p.ucomp[i];
                    const long long tg = p.colg[t / R];
                    min_all = tg < min_all ? tg : min_all;
                    if (p.agg_flag[j])
                    {
                        p.t_finished[t] = 1;
                        p.t_cid[t] = p.agg_cid[j];
                    }
                    else
                    {
                        surv = true;
                        min_surv = tg < min_surv ? tg : min_surv;
                        if (j == i)
                        {
                            const double fin = __longlong_as_double((long long) ld_agent(&p.agg_fin[i]));
                            L_new = fin < L_new ? fin : L_new;
                        }
                    }
                }
                const unsigned long long mask = __ballot(surv);
                if (surv)
                {
                    const int np = out + __popcll(mask & lanes_below());
                    p.ulist[np] = t;
                    p.t_pos[t] = np;
                }
                out += __popcll(mask);
            }
            min_all = wave_min_i64(min_all);
            min_surv = wave_min_i64(min_surv);
            L = wave_min_f64(L_new);
            M_c = min_all;
            M = min_surv;
            n_unf = out;
            assoc_wave_sync();
        }
        last_min_az = min_az;

        // ------------------------------------------------------------------ publish + clear (cc.cpp:1035-1145)
        if (M_c < first_unpub)
        {
            err = CC_ERR_BOOKKEEPING;
            err_a = M_c;
            err_b = first_unpub;
            break;
        }
        const long long old_unpub = first_unpub, old_ring = ring_start;
        first_unpub = M_c;
        ring_start = first_unpub - NC > 0 ? first_unpub - NC : 0;
        emit(CC_EV_PUBLISH_COLUMNS, old_unpub, first_unpub - 1, 0, 0, gc);
        // cluster ids of the published cells are written by k_publish after this kernel
        cells_published += (unsigned long long) (first_unpub - old_unpub) * (unsigned long long) R;
        // physical clearing of [old_ring, ring_start) is deferred to the next k_insert (StreamState::clear_done)
        (void) old_ring;
    }

    if (lane == 0)
    {
        st->first_unpublished = first_unpub;
        st->batch[slot].pub_end = first_unpub;
        st->ring_start = ring_start;
        st->cluster_counter = cluster_counter;
        st->n_unfinished = n_unf;
        st->min_required = M;
        st->finish_lower_bound = L;
        st->last_round_min_az = last_min_az;
        st->cells_published = cells_published;
        st->clusters_finished = clusters_finished;
        st->exceed_one_rotation = exceed;
        st->serial_columns = serial_cols;
        st->stamp_alias_rounds = alias_rounds;
        st->batch[slot].acp_next = col_end;
        // back to the LDS kernels once the unfinished trees fit their pool comfortably again (never for window configurations
        // they do not support)
        if (err == 0)
            st->assoc_mode = (cfg.max_steps_in_row > WIN_COLS - 2 || n_unf * 2 > g.lds_tree_limit) ? 1 : 0;
        st->n_events = n_events < g.event_capacity ? n_events : g.event_capacity;
        if (g.record_events && n_events > g.event_capacity && err == 0)
        {
            err = CC_ERR_CAPACITY;
            err_a = n_events;
        }
        if (err)
            raise_error(st, err, err_a, err_b);
    }
}

template<int RPL>
__global__ __launch_bounds__(64) void k_associate(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot)
{
    associate_stream<RPL>(g, cfg, P, states, first_stream + (int) blockIdx.x, slot);
}


// ---- column epilogue of the window scan: everything about the column that does not depend on the tree state, so that the serial
// association kernel finds it precomputed. (1) where every point's chain of same-column parents ends; (2) the column summary. One
// wavefront, lanes = rows; `parent` = (columns back << 8) | row of the first accepted candidate, -1 new root, -2 ignored cell.
template<int RPL, bool MIRROR>
__device__ __forceinline__ void scan_column_epilogue(const SP& p, const int R, const int lc, const int lane, const int (&parent)[RPL],
                                                     const int (&nlinks)[RPL], const double (&fin)[RPL], const unsigned long long (&packed)[RPL],
                                                     int reach)
{
    int t[RPL]; // row at the top of the chain so far
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        const int row = k * 64 + lane;
        const bool same_col = parent[k] >= 0 && (parent[k] >> 8) == 0;
        t[k] = same_col ? (parent[k] & 0xff) : row;
    }
    for (int it = 0; it < 7; it++) // pointer jumping: rows <= 128, chains shorter than 2^7
    {
        int nt[RPL];
        bool changed = false;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int src = t[k];
            const int lo = __shfl(t[0], src & 63);
            const int hi = RPL > 1 ? __shfl(t[RPL - 1], src & 63) : lo;
            nt[k] = src < 64 ? lo : hi;
            changed |= nt[k] != t[k];
        }
#pragma unroll
        for (int k = 0; k < RPL; k++)
            t[k] = nt[k];
        if (!__any(changed))
            break;
    }
    int cnt_new = 0, mine[RPL];
    int max_delta = 0;
    int flags = 0;
    int n_act = 0; // active points of the column, 8 bits per 64 rows
    double newfin = 1.7976931348623157e308;
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        n_act |= __popcll(__ballot(parent[k] >= -1)) << (8 * k);
        const bool is_new = parent[k] == -1;
        const unsigned long long mask = __ballot(is_new);
        const int newidx = cnt_new + __popcll(mask & lanes_below());
        cnt_new += __popcll(mask);
        mine[k] = is_new ? newidx : (parent[k] >= 0 ? parent[k] : -1);
        if (is_new && fin[k] < newfin)
            newfin = fin[k];
        if (parent[k] >= 0)
        {
            int d = parent[k] >> 8;
            const int nl = nlinks[k] == 255 ? LINK_SLOTS : nlinks[k];
            for (int j = 0; j < nl; j++)
            {
                const int dj = (int) ((packed[k] >> (16 * j + 8)) & 0xff);
                d = dj > d ? dj : d;
            }
            max_delta = d > max_delta ? d : max_delta;
        }
        if (nlinks[k] == 255)
            flags |= 1;
        else if (nlinks[k] > 0)
            flags |= 2;
    }
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        const int row = k * 64 + lane;
        const int src = t[k];
        const int lo = __shfl(mine[0], src & 63);
        const int hi = RPL > 1 ? __shfl(mine[RPL - 1], src & 63) : lo;
        const int term = parent[k] < -1 ? -1 : (src < 64 ? lo : hi);
        if (row < R)
            p.sc_term[lc * R + row] = (int16_t) term;
        // what the batch-parallel association reads: the column's ACTIVE points packed in row order (entry j of the column at lc * R + j), so that
        // it works on full lanes; the cells without a point get their (absent) tree root here
        const unsigned long long actm = __ballot(parent[k] >= -1);
        if (parent[k] >= -1)
        {
            const int j = (k > 0 ? (n_act & 0xff) : 0) + __popcll(actm & lanes_below());
            const unsigned nlc = nlinks[k] == 255 ? 7u : (unsigned) nlinks[k];
            // sc_term (16 bits) | row << 16 | link count (0 .. 4, 7 = overflowed) << 23 | new root << 26
            p.pk_meta[lc * R + j] = ((unsigned) term & 0xffffu) | ((unsigned) row << 16) | (nlc << 23) | (parent[k] == -1 ? 1u << 26 : 0u);
            p.pk_fin[lc * R + j] = fin[k];
            if (nlinks[k] > 0)
                p.pk_lk[lc * R + j] = packed[k];
        }
        else if (row < R)
            p.root[lc * R + row] = -1;
    }
    max_delta = -wave_min_i32(-max_delta); // DPP reductions, ballots: no LDS round trips
    if (MIRROR)
        reach = -wave_min_i32(-reach);
    flags = (__any(flags & 1) ? 1 : 0) | (__any(flags & 2) ? 2 : 0);
    if (cnt_new > 0) // (wave-uniform; four columns of five have no new root, and the 64-bit reduction is ~30 instructions)
        newfin = wave_min_f64(newfin);
    if (lane == 0)
    {
        p.col_newfin[lc] = newfin;
        p.col_info[lc] = cnt_new | (flags << 8) | (max_delta << 16) | ((MIRROR ? reach : 0) << 24);
        p.col_act[lc] = (uint16_t) n_act;
    }
}

// =====================================================================================================
// k_scan — the window scan of traverseFieldOfView (cc.cpp:698-771) for every point of the batch's columns, as a pure
// function of static per-cell data (SURVEY.md 8a "derived fact"): first accepted candidate = parent, later accepted
// candidates = links, early stops as if the first match roots the point. Massively parallel; the serial kernel below
// validates the assumption per column (no refused attach, nothing used from columns the live scan would not reach).
// grid = (streams, SCAN_BLOCKS), block = 64, blocks stride over the columns of the batch. The stream index is the fast grid
// dimension: workgroups are dealt to the 8 XCDs round-robin by linear id, so with a multiple of 8 streams all blocks of one stream
// run on one XCD and share its L2 (every candidate column is read by the scans of several later columns).
// =====================================================================================================
#ifndef CC_SCAN_BLOCKS
#define CC_SCAN_BLOCKS 256
#endif
constexpr int SCAN_BLOCKS = CC_SCAN_BLOCKS;

// MIRROR: also count Point::number_of_visited_neighbors (cc.cpp:725) and how far back the scan looked (the live scan stops at the first
// unpublished column, cc.cpp:762-763, so a count is only right if it did not look past it: the association kernels replay such columns).
// (a device function: k_scan is its kernel — grid (streams, SCAN_BLOCKS), one wavefront per block —; k_small_front runs it on its four wavefronts)
template<int RPL, bool MIRROR>
__device__ __forceinline__ void scan_body(const Geometry& g, const cc_config& cfg, const Planes& P, StreamState* states, int first_stream, int slot,
                                          const int bx, const int by, const int ny)
{
    const int s = first_stream + bx;
    const int lane = lane_id();
    const StreamState* st = &states[s];
    if (st->error != 0 || st->batch[slot].seg_begin < 0 || st->batch[slot].mode != 0)
        return;
    AssocCtx c;
    c.p = stream_ptrs(P, g, s);
    const SP& p = c.p;
    const int R = c.R = g.num_rows;
    c.NC = g.num_columns;
    const int RC = c.RC = g.ring_cols;
    c.az_width = g.az_width;
    c.maxd2 = g.max_distance_squared;
    c.max_steps_in_row = cfg.max_steps_in_row;
    c.max_steps_in_column = cfg.max_steps_in_column;
    c.stop_enabled = cfg.stop_after_association_enabled;
    c.stop_min_steps = cfg.stop_after_association_min_steps;
    const long long col_end = st->batch[slot].seg_end, first_column = st->first_column;
    // (ring columns advanced incrementally: a 64-bit modulo per column costs ~100 scalar instructions)
    const int first_lc = (int) (first_column % RC);
    int lc = (int) ((st->batch[slot].acp_next + by) % RC);
    const int lc_step = (int) ((unsigned) ny % (unsigned) RC);
    const int NC = g.num_columns;
    long long rot = (st->batch[slot].acp_next + by) / NC; // rotation index / column within the rotation, advanced the same way
    int cir = (int) ((st->batch[slot].acp_next + by) - rot * NC);
    const int cir_step = (int) ((unsigned) ny % (unsigned) NC);
    const long long rot_step = (long long) ((unsigned) ny / (unsigned) NC);
    CazBase cb = caz_base_of_rotation(rot);
    long long cb_rot = rot;
    for (long long gc = st->batch[slot].acp_next + by; gc < col_end;
         gc += (unsigned) ny, lc = (lc + lc_step >= RC ? lc + lc_step - RC : lc + lc_step), rot += rot_step + (cir + cir_step >= NC ? 1 : 0),
                   cir = (cir + cir_step >= NC ? cir + cir_step - NC : cir + cir_step))
    {
        // never look at columns older than the first column ever segmented (their planes are uninitialised)
        const int bound = (gc - first_column) <= (long long) cfg.max_steps_in_row + 1 ? first_lc : -1;
        if (rot != cb_rot)
        {
            cb = caz_base_of_rotation(rot);
            cb_rot = rot;
        }
        int parent[RPL], nlinks[RPL];
        double fin[RPL];
        unsigned long long packed[RPL];
        int reach = 0; // deepest column (steps back) any visit of this lane went to
        if constexpr (RPL == 1)
        {
            // Rows = lanes: the scan of all 64 points of the column runs in lock step. Every lane visits the same relative cell
            // (sb columns back, d rows up or down) at the same time, in the reference's order (cc.cpp:706-769): the candidate
            // column is loaded once per sb (one coalesced 16-byte record per lane) and the cell each lane wants arrives by a
            // cross-lane read, instead of one gathered load plus divergent-loop bookkeeping per visit and lane.
            const int row = lane;
            const int ci = lc * R + row;
            const bool inrow = row < R;
            float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
            bool live = false; // this lane's point is still scanning further columns
            float mad = 0.f;
            int needed = -1;
            parent[0] = -2;
            nlinks[0] = 0;
            fin[0] = 0.;
            packed[0] = 0;
            if (inrow && !p.ignored[ci])
            {
                live = true;
                parent[0] = -1;
                me = p.sc_rec[ci]; // the point itself is not ignored: x is the real coordinate
                mad = ccm::asinf_exact(cfg.max_distance / p.dist[ci]);
                fin[0] = cell_caz(cb, p.incaz[ci]) + (double) mad;
                needed = f2i_x86(__builtin_ceilf(mad / c.az_width));
                needed = needed < c.max_steps_in_row ? needed : c.max_steps_in_row;
            }
            // per-lane state as 0/1 integers in VGPRs: booleans carried through the loops as lane masks cost three scalar
            // instructions per variable at every loop exit
            int rooted = 0, overflow = 0, live_i = live ? 1 : 0, visits = 0;
            int oc = lc;
            for (int sb = 0;; sb++)
            {
                live_i = (live_i && sb <= needed) ? 1 : 0;
                if (!__any(live_i != 0))
                    break;
                float4 cr = make_float4(0.f, 0.f, 0.f, 0.f);
                if (inrow)
                {
                    cr = p.sc_rec[oc * R + row];
                    if (p.ignored[oc * R + row])
                        cr.x = __builtin_nanf(""); // in registers only: an ignored cell travels through the cross-lane reads as x = NaN
                }
                for (int down = 0; down < 2; down++) // dir = -1 (rows above), then dir = +1 (cc.cpp:712-716)
                {
                    if (down == 1 && sb == 0)
                        continue;
                    int d = (down == 1 || sb == 0) ? 1 : 0; // d = sv = |orow - row|
                    int orow = down ? row + d : row - d;
                    int run = (live_i && orow >= 0 && orow < R && d <= c.max_steps_in_column) ? 1 : 0;
                    while (__any(run != 0))
                    {
                        const int src = orow & 63;
                        const float ox = __shfl(cr.x, src), oy = __shfl(cr.y, src), oz = __shfl(cr.z, src), ow = __shfl(cr.w, src);
                        // branch-free: cc.cpp:721 inclination window, :729 ignored cell, :738 distance, :745-757 parent / link,
                        // :759 early stop
                        if (MIRROR)
                        {
                            visits += run; // cc.cpp:725
                            reach = run ? sb : reach;
                        }
                        const int cont = (run && !(ccm::absf(ow - me.w) > mad)) ? 1 : 0;
                        const float dx = me.x - ox, dy = me.y - oy, dz = me.z - oz;
                        const int acc = (cont && ox == ox && dx * dx + dy * dy + dz * dz < c.maxd2) ? 1 : 0; // x = NaN: ignored / empty
                        const int cand = (sb << 8) | (orow & 0xff);
                        parent[0] = (acc && !rooted) ? cand : parent[0];
                        if (__any(acc && rooted)) // a second accepted candidate is a link (rare next to the visits: wave-uniform branch)
                        {
                            const int as_link = (acc && rooted && nlinks[0] < LINK_SLOTS) ? 1 : 0;
                            overflow |= (acc && rooted && nlinks[0] >= LINK_SLOTS) ? 1 : 0;
                            packed[0] |= as_link ? (unsigned long long) cand << (16 * nlinks[0]) : 0ull;
                            nlinks[0] += as_link;
                        }
                        rooted |= acc;
                        const int stop = (rooted && c.stop_enabled && d >= c.stop_min_steps) ? 1 : 0;
                        d++;
                        orow = down ? orow + 1 : orow - 1;
                        run = (cont && !stop && orow >= 0 && orow < R && d <= c.max_steps_in_column) ? 1 : 0;
                    }
                }
                if (rooted && c.stop_enabled && sb >= c.stop_min_steps)
                    live_i = 0;
                if (oc == bound)
                    break;
                oc = oc == 0 ? RC - 1 : oc - 1;
            }
            if (overflow)
                nlinks[0] = 255;
            if (inrow)
            {
                p.sc_parent[ci] = (int16_t) parent[0];
                p.sc_nlinks[ci] = (uint8_t) nlinks[0];
                p.sc_fin[ci] = fin[0];
                if (nlinks[0] > 0)
                    p.sc_links[ci] = packed[0];
                if (MIRROR)
                    p.sc_visits[ci] = sat_u16(visits);
            }
        }
        else
        {
            static_assert(RPL == 2, "one or two rows per lane");
            // Two rows per lane (65 - 128 rows), the same lock step: both of a lane's points visit the same relative cell at the same time. The
            // candidate column is two coalesced records per lane (rows lane and 64 + lane); the cell row - d of the upper half lies in the upper
            // half's registers of lane - d, that of the lower half in the lower half's registers of lane - d — or, for the first d lanes, in
            // the upper half's of lane - d + 64 (mod 64 the same lane): two cross-lane reads per component and one select for the half that
            // crosses. Round 4: the per-lane gathers of k_scan2 kept the texture addresser busy 64 clocks per visit (1.69 ms alone at 256 x S128).
            float4 me[2];
            float mad[2];
            int needed[2], rooted[2], overflow[2], live_i[2], visits[2], reachk[2];
            bool inrow[2];
#pragma unroll
            for (int k = 0; k < 2; k++)
            {
                const int row = k * 64 + lane;
                const int ci = lc * R + (row < R ? row : 0);
                inrow[k] = row < R;
                me[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                mad[k] = 0.f;
                needed[k] = -1;
                parent[k] = -2;
                nlinks[k] = 0;
                fin[k] = 0.;
                packed[k] = 0;
                rooted[k] = overflow[k] = live_i[k] = visits[k] = reachk[k] = 0;
                if (inrow[k] && !p.ignored[ci])
                {
                    live_i[k] = 1;
                    parent[k] = -1;
                    me[k] = p.sc_rec[ci];
                    mad[k] = ccm::asinf_exact(cfg.max_distance / p.dist[ci]);
                    fin[k] = cell_caz(cb, p.incaz[ci]) + (double) mad[k];
                    needed[k] = f2i_x86(__builtin_ceilf(mad[k] / c.az_width));
                    needed[k] = needed[k] < c.max_steps_in_row ? needed[k] : c.max_steps_in_row;
                }
            }
            int oc = lc;
            for (int sb = 0;; sb++)
            {
#pragma unroll
                for (int k = 0; k < 2; k++)
                    live_i[k] = (live_i[k] && sb <= needed[k]) ? 1 : 0;
                if (!__any((live_i[0] | live_i[1]) != 0))
                    break;
                float4 cr[2];
#pragma unroll
                for (int k = 0; k < 2; k++)
                {
                    cr[k] = make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
                    if (inrow[k])
                    {
                        cr[k] = p.sc_rec[oc * R + k * 64 + lane];
                        if (p.ignored[oc * R + k * 64 + lane])
                            cr[k].x = __builtin_nanf("");
                    }
                }
                for (int down = 0; down < 2; down++) // dir = -1 (rows above), then dir = +1 (cc.cpp:712-716)
                {
                    if (down == 1 && sb == 0)
                        continue;
                    int d = (down == 1 || sb == 0) ? 1 : 0;
                    int run[2];
#pragma unroll
                    for (int k = 0; k < 2; k++)
                    {
                        const int orow = down ? k * 64 + lane + d : k * 64 + lane - d;
                        run[k] = (live_i[k] && orow >= 0 && orow < R && d <= c.max_steps_in_column) ? 1 : 0;
                    }
                    while (__any((run[0] | run[1]) != 0))
                    {
                        const int src = (down ? lane + d : lane - d) & 63;
                        const float a0x = __shfl(cr[0].x, src), a0y = __shfl(cr[0].y, src), a0z = __shfl(cr[0].z, src), a0w = __shfl(cr[0].w, src);
                        const float a1x = __shfl(cr[1].x, src), a1y = __shfl(cr[1].y, src), a1z = __shfl(cr[1].z, src), a1w = __shfl(cr[1].w, src);
                        // (the wanted row k * 64 + lane -/+ d lies in lane (lane -/+ d) mod 64 of the half its bit 6 names)
#pragma unroll
                        for (int k = 0; k < 2; k++)
                        {
                            const int orow = down ? k * 64 + lane + d : k * 64 + lane - d;
                            const int half = orow >> 6; // 0 or 1 where the visit is wanted (run[k]); anything else is not used
                            const bool h1 = half == 1;
                            const float ox = h1 ? a1x : a0x, oy = h1 ? a1y : a0y, oz = h1 ? a1z : a0z, ow = h1 ? a1w : a0w;
                            if (MIRROR)
                            {
                                visits[k] += run[k]; // cc.cpp:725
                                reachk[k] = run[k] ? sb : reachk[k];
                            }
                            const int cont = (run[k] && !(ccm::absf(ow - me[k].w) > mad[k])) ? 1 : 0;
                            const float dx = me[k].x - ox, dy = me[k].y - oy, dz = me[k].z - oz;
                            const int acc = (cont && ox == ox && dx * dx + dy * dy + dz * dz < c.maxd2) ? 1 : 0; // x = NaN: ignored / empty
                            const int cand = (sb << 8) | (orow & 0xff);
                            parent[k] = (acc && !rooted[k]) ? cand : parent[k];
                            if (__any(acc && rooted[k])) // a second accepted candidate is a link (rare next to the visits: wave-uniform branch)
                            {
                                const int as_link = (acc && rooted[k] && nlinks[k] < LINK_SLOTS) ? 1 : 0;
                                overflow[k] |= (acc && rooted[k] && nlinks[k] >= LINK_SLOTS) ? 1 : 0;
                                packed[k] |= as_link ? (unsigned long long) cand << (16 * nlinks[k]) : 0ull;
                                nlinks[k] += as_link;
                            }
                            rooted[k] |= acc;
                            const int stop = (rooted[k] && c.stop_enabled && d >= c.stop_min_steps) ? 1 : 0;
                            const int nrow = down ? orow + 1 : orow - 1;
                            run[k] = (cont && !stop && nrow >= 0 && nrow < R && d + 1 <= c.max_steps_in_column) ? 1 : 0;
                        }
                        d++;
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; k++)
                    if (rooted[k] && c.stop_enabled && sb >= c.stop_min_steps)
                        live_i[k] = 0;
                if (oc == bound)
                    break;
                oc = oc == 0 ? RC - 1 : oc - 1;
            }
#pragma unroll
            for (int k = 0; k < 2; k++)
            {
                const int row = k * 64 + lane;
                if (overflow[k])
                    nlinks[k] = 255;
                if (inrow[k])
                {
                    const int ci = lc * R + row;
                    p.sc_parent[ci] = (int16_t) parent[k];
                    p.sc_nlinks[ci] = (uint8_t) nlinks[k];
                    p.sc_fin[ci] = fin[k];
                    if (nlinks[k] > 0)
                        p.sc_links[ci] = packed[k];
                    if (MIRROR)
                        p.sc_visits[ci] = sat_u16(visits[k]);
                }
                if (MIRROR)
                    reach = reachk[k] > reach ? reachk[k] : reach;
            }
        }
        scan_column_epilogue<RPL, MIRROR>(p, R, lc, lane, parent, nlinks, fin, packed, reach);
    }
}

template<int RPL, bool MIRROR>
__global__ __launch_bounds__(64) void k_scan(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot)
{
    scan_body<RPL, MIRROR>(g, cfg, P, states, first_stream, slot, (int) blockIdx.x, (int) blockIdx.y, (int) gridDim.y);
}

// =====================================================================================================
// k_small_front — everything up to and including the window scan for a call of a few firings on ONE stream, in one launch (the per-column latency
// path, cc_engine_add_firings with n < 64): what k_begin_batch, k_ego, k_prep, k_insert2 and k_seg_small do one after the other. A captured
// graph spends ~4.5 us per kernel node on a call whose kernels need 1 - 10 us each; five nodes less are ~20 us of a 65 us call.
// grid = 1, block = 256, dynamic LDS = insert2_lds_bytes(num_rows); num_rows <= 64.
//   A  all threads: the batch begins (thread 0), per-firing ego records, per-point preparation into the staging planes
//   B  wavefronts 0 and 1: the serial insertion (insert2_body: consumer + loader)
//   C  wavefront 0: the segmentation of the columns the call finished (seg_small_body)
//   D  all wavefronts: the window scan of those columns (scan_body)
// =====================================================================================================
__global__ __launch_bounds__(256) void k_small_front(Geometry g, cc_config cfg, Planes P, StreamState* states, int stream, int slot,
                                                     const float* __restrict__ xyz, const uint8_t* __restrict__ inten, const double* __restrict__ poses,
                                                     long long n, int* remaining, double* __restrict__ ego)
{
    const int R = g.num_rows;
    StreamState* st = &states[stream];
#ifdef CC_SF_STATS
    unsigned long long sf_t[6];
    sf_t[0] = __builtin_amdgcn_s_memtime();
#define SF_MARK(i) sf_t[i] = __builtin_amdgcn_s_memtime();
#else
#define SF_MARK(i)
#endif
    if (threadIdx.x == 0)
    {
        // k_begin_batch (cc_engine.hip) for this stream; a call on the host path never clears past what the host has seen
        st->cursor = 0;
        st->par_bad = 0x7fffffff;
        st->par_upto = -1;
        st->par_clear_done = -1;
        st->pre_seg_begin = 0;
        st->n_events = 0;
        st->n_links = 0;
        st->batch[slot].fused = 0;
        st->clear_allowed = st->ring_start;
        *remaining = 0;
    }
    for (long long f = threadIdx.x; f < n; f += 256)
        ego_record(states, stream, cfg, poses, n, n, 0, ego, 0, f);
    for (long long i = threadIdx.x; i < n * R; i += 256)
    {
        const PreppedPoint q = prep_point(xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], poses + (i / R) * 12, cfg.sensor_is_clockwise != 0, g.az_width);
        P.pp_cir[i] = q.cir;
        if (q.cir == PP_SKIP)
            continue;
        P.pp_x[i] = q.x;
        P.pp_y[i] = q.y;
        P.pp_z[i] = q.z;
        P.pp_dist[i] = q.dist;
        P.pp_incl[i] = q.incl;
        P.pp_incaz[i] = q.incaz;
    }
    __syncthreads(); // (workgroup-scope release / acquire: the staging planes, the ego records and the stream state are visible to wavefronts 0 and 1)
    SF_MARK(1)
    if (threadIdx.x < 128)
        insert2_body<1, true>(g, cfg, P, states, stream, slot, inten, n, remaining, n, 0, 0);
    else
        __syncthreads(); // (insert2_body has ONE block barrier, right at its start: the wavefronts that do not run it must meet it, or every
                         // barrier behind it pairs the wrong phases — the hardware only counts arrivals)
    __syncthreads();
    SF_MARK(2)
    if (threadIdx.x < 64)
        seg_small_body(g, cfg, P, states, stream, slot, poses, n, 0, ego, n, 0);
    __syncthreads();
    SF_MARK(3)
    // D  all four wavefronts: the window scan of the call's columns (scan_body: what k_scan does with one wavefront per block)
    if (g.mirror_fields)
        scan_body<1, true>(g, cfg, P, states, stream, slot, 0, uniform_i32((int) (threadIdx.x >> 6)), 4);
    else
        scan_body<1, false>(g, cfg, P, states, stream, slot, 0, uniform_i32((int) (threadIdx.x >> 6)), 4);
#ifdef CC_SF_STATS
    __syncthreads();
    SF_MARK(4)
    if (threadIdx.x == 0)
    {
        for (int i = 0; i < 4; i++)
            st->dbg[i] += sf_t[i + 1] - sf_t[i];
        st->dbg[4] += 1;
    }
#endif
}

// =====================================================================================================
// k_scan2 — the same window scan with the ACTIVE points packed into the lanes. Only every fourth cell reaches the association (ground,
// ego, empty and filtered cells are ignored) and nearly every scan is over after four visits (cc.cpp:746-758), so a wavefront whose
// lanes are the rows of one column runs its lock-step visit loop for the slowest of ~16 busy lanes while 48 idle ones ride along. Here a
// wavefront takes a tile of SCAN_TILE_CELLS / num_rows columns, compacts the non-ignored cells of the tile into a list, and every lane
// scans ONE point of the list with its own little state machine (one visit per iteration, the candidate's 16-byte record by a gather
// that hits L2: k_seg_pre / k_seg_scan wrote the records just before). Results go through LDS back into rows-as-lanes order for the
// column epilogue (same-column parent chains, column summary) and the coalesced stores. Same outputs as k_scan, bit for bit.
// grid = (streams, SCAN_BLOCKS), block = 64.
// =====================================================================================================
// (256 cells: ~64 active points = one packed pass. 512 — fuller passes, half the tiles — is 15 % slower at 64 rows and 6 % at 128: twice the
// LDS per one-wavefront block and longer tails of the per-lane state machines)
#ifndef CC_SCAN_TILE_CELLS
#define CC_SCAN_TILE_CELLS 256
#endif
constexpr int SCAN_TILE_CELLS = CC_SCAN_TILE_CELLS;

template<int RPL, bool MIRROR>
__global__ __launch_bounds__(64) void k_scan2(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot)
{
    const int s = first_stream + blockIdx.x;
    const int lane = lane_id();
    const StreamState* st = &states[s];
    if (st->error != 0 || st->batch[slot].seg_begin < 0 || st->batch[slot].mode != 0)
        return;
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols;
    const float maxd2 = g.max_distance_squared;
    const int max_row_steps = cfg.max_steps_in_row, max_col_steps = cfg.max_steps_in_column;
    const bool stop_enabled = cfg.stop_after_association_enabled != 0;
    const int stop_min = cfg.stop_after_association_min_steps;
    const int TC = SCAN_TILE_CELLS / (RPL * 64); // columns per tile: 4 at <= 64 rows, 2 at <= 128
    __shared__ unsigned short s_list[SCAN_TILE_CELLS];  // tile-local cell (column in tile * RPL * 64 + row) of every active point
    __shared__ short s_parent[SCAN_TILE_CELLS];
    __shared__ unsigned char s_nlinks[SCAN_TILE_CELLS];
    __shared__ unsigned short s_visits[SCAN_TILE_CELLS];
    __shared__ unsigned char s_reach[SCAN_TILE_CELLS];
    __shared__ double s_fin[SCAN_TILE_CELLS];
    __shared__ unsigned long long s_links[SCAN_TILE_CELLS];
    const long long col_begin = st->batch[slot].acp_next, col_end = st->batch[slot].seg_end, first_column = st->first_column;
    const int first_lc = (int) (first_column % RC);
    const long long n_tiles = (col_end - col_begin + TC - 1) / TC;
    for (long long tile = blockIdx.y; tile < n_tiles; tile += gridDim.y)
    {
        const long long gc0 = col_begin + tile * TC;
        const int ncols = (int) (col_end - gc0 < TC ? col_end - gc0 : TC);
        const int lc0 = (int) (gc0 % RC);
        const long long rot0 = gc0 / g.num_columns; // rotation index and column within the rotation of the tile's first column
        const int cir0 = (int) (gc0 - rot0 * g.num_columns);
        // ---- A: the tile's active cells --------------------------------------------------------------------------------------
        int n_act = 0;
        for (int j = 0; j < TC * RPL; j++)
        {
            const int tc = j / RPL, row = (j % RPL) * 64 + lane;
            const int tl = tc * RPL * 64 + row;
            bool act = false;
            if (tc < ncols && row < R)
            {
                int lc = lc0 + tc;
                lc = lc >= RC ? lc - RC : lc;
                act = p.ignored[lc * R + row] == 0;
            }
            s_parent[tl] = -2;
            s_nlinks[tl] = 0;
            s_fin[tl] = 0.;
            s_links[tl] = 0;
            if (MIRROR)
            {
                s_visits[tl] = 0;
                s_reach[tl] = 0;
            }
            const unsigned long long m = __ballot(act);
            if (act)
                s_list[n_act + __popcll(m & lanes_below())] = (unsigned short) tl;
            n_act += __popcll(m);
        }
        wave_lds_fence();
        // ---- B: one point per lane ---------------------------------------------------------------------------------------------
        for (int base = 0; base < n_act; base += 64)
        {
            const bool have = base + lane < n_act;
            const int tl = have ? (int) s_list[base + lane] : 0;
            const int tc = tl / (RPL * 64), row = tl % (RPL * 64);
            const long long gc = gc0 + tc;
            int lc = lc0 + tc;
            lc = lc >= RC ? lc - RC : lc;
            const int ci = lc * R + row;
            float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
            float mad = 0.f;
            double fin = 0.;
            int needed = -1;
            if (have)
            {
                me = p.sc_rec[ci];
                mad = ccm::asinf_exact(cfg.max_distance / p.dist[ci]);
                fin = cell_caz(caz_base_of_rotation(rot0 + (cir0 + tc >= g.num_columns ? 1 : 0)), p.incaz[ci]) + (double) mad;
                needed = f2i_x86(__builtin_ceilf(mad / g.az_width));
                needed = needed < max_row_steps ? needed : max_row_steps;
            }
            // never look at columns older than the first column ever segmented (their planes are uninitialised)
            const int bound = (gc - first_column) <= (long long) max_row_steps + 1 ? first_lc : -1;
            // state of the scan (cc.cpp:706-769): column offset sb, direction (0 = rows above, 1 = rows below), vertical step d
            int sb = 0, down = 0, d = 1, oc = lc, orow = row - 1;
            int rooted = 0, parent = -1, nlinks = 0, overflow = 0, visits = 0, reach = 0;
            unsigned long long packed = 0;
            // position on the first cell that passes the while-condition of cc.cpp:718-719, or finish
            bool run = have;
            auto next_column = [&]() // the end of a column's visits: cc.cpp:756-769
            {
                if ((rooted && stop_enabled && sb >= stop_min) || oc == bound || sb + 1 > needed)
                    run = false;
                else
                {
                    sb++;
                    oc = oc == 0 ? RC - 1 : oc - 1;
                    down = 0;
                    d = 0;
                    orow = row; // (the cell in the same row always passes the loop condition: d = 0, row inside the image)
                }
            };
            auto next_direction = [&]() // a direction ended (break or loop condition false)
            {
                if (down == 0 && sb > 0)
                {
                    down = 1;
                    d = 1;
                    orow = row + 1;
                    if (!(orow < R && d <= max_col_steps))
                        next_column();
                }
                else
                    next_column();
            };
            if (run && !(orow >= 0 && d <= max_col_steps))
                next_direction(); // row 0 has nothing above it in its own column
            while (__any(run))
            {
                if (run)
                {
                    const float4 o = p.sc_rec[oc * R + orow];
                    const unsigned char oign = p.ignored[oc * R + orow]; // (issued with the record: one round trip per visit)
                    if (MIRROR)
                    {
                        visits++; // cc.cpp:725
                        reach = sb;
                    }
                    if (ccm::absf(o.w - me.w) > mad) // cc.cpp:728: the inclination window is left
                        next_direction();
                    else
                    {
                        const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;
                        if (!oign && dx * dx + dy * dy + dz * dz < maxd2) // (a cell without a return is ignored, and its x is NaN)
                        {
                            const int cand = (sb << 8) | orow;
                            if (!rooted)
                                parent = cand;
                            else if (nlinks < LINK_SLOTS)
                            {
                                packed |= (unsigned long long) cand << (16 * nlinks);
                                nlinks++;
                            }
                            else
                                overflow = 1;
                            rooted = 1;
                        }
                        if (rooted && stop_enabled && d >= stop_min) // cc.cpp:746-749
                            next_direction();
                        else
                        {
                            d++;
                            orow = down ? orow + 1 : orow - 1;
                            if (!(orow >= 0 && orow < R && d <= max_col_steps))
                                next_direction();
                        }
                    }
                }
            }
            if (have)
            {
                s_parent[tl] = (short) parent;
                s_nlinks[tl] = (unsigned char) (overflow ? 255 : nlinks);
                s_fin[tl] = fin;
                s_links[tl] = packed;
                if (MIRROR)
                {
                    s_visits[tl] = sat_u16(visits);
                    s_reach[tl] = (unsigned char) reach;
                }
            }
        }
        wave_lds_fence();
        // ---- C: back to rows-as-lanes: stores and the column epilogue ------------------------------------------------------------
        for (int tc = 0; tc < ncols; tc++)
        {
            int lc = lc0 + tc;
            lc = lc >= RC ? lc - RC : lc;
            int parent[RPL], nlinks[RPL];
            double fin[RPL];
            unsigned long long packed[RPL];
            int reach = 0;
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                const int tl = tc * RPL * 64 + row;
                parent[k] = s_parent[tl];
                nlinks[k] = s_nlinks[tl];
                fin[k] = s_fin[tl];
                packed[k] = s_links[tl];
                if (MIRROR)
                    reach = (int) s_reach[tl] > reach ? (int) s_reach[tl] : reach;
                if (row < R)
                {
                    const int ci = lc * R + row;
                    p.sc_parent[ci] = (int16_t) parent[k];
                    p.sc_nlinks[ci] = (uint8_t) nlinks[k];
                    p.sc_fin[ci] = fin[k];
                    if (nlinks[k] > 0)
                        p.sc_links[ci] = packed[k];
                    if (MIRROR)
                        p.sc_visits[ci] = s_visits[tl];
                }
            }
            scan_column_epilogue<RPL, MIRROR>(p, R, lc, lane, parent, nlinks, fin, packed, reach);
        }
        wave_lds_fence(); // the tile's LDS arrays are rewritten by the next tile
    }
}

// =====================================================================================================
// k_assoc_lds — association bookkeeping, union-find, finished-cluster check and publishing (cc.cpp:643-696, 773-1092)
// with every hot structure in LDS: tree-slot ids of the last WIN_COLS columns, the unfinished point trees
// (sc_unfinished_point_trees_) as dense slot arrays in creation order, and the per-cluster aggregates of the finish
// check. One wavefront per stream, lanes = rows, serial over the columns of the batch; consumes k_scan's staging.
// =====================================================================================================
struct LdsTrees
{
    int cell[TREE_SLOTS];                // root cell of the tree in list position i
    long long gcol[TREE_SLOTS];          // its global column
    unsigned long long fin[TREE_SLOTS];  // bits of finished_at_continuous_azimuth_angle (non-negative double)
    unsigned last[TREE_SLOTS];           // low 32 bits of the last global column that attached a point (width = last - gcol + 1)
    unsigned pts[TREE_SLOTS];
    int uf[TREE_SLOTS];                  // union-find parent (list position)
    unsigned long long c_fin[TREE_SLOTS]; // at a representative: lower bound of the cluster's max finished_at (exact after a scan)
    // finish check scratch
    unsigned long long a_fin[TREE_SLOTS];
    long long a_min[TREE_SLOTS];
    long long a_max[TREE_SLOTS];
    unsigned a_pts[TREE_SLOTS];
    int a_first[TREE_SLOTS];
    unsigned a_cid[TREE_SLOTS];
    int comp[TREE_SLOTS];
    int remap[TREE_SLOTS];
    unsigned char a_flag[TREE_SLOTS];
};

__device__ __forceinline__ int lds_find(int* uf, int a)
{
    while (true)
    {
        const int pa = lds_ld(&uf[a]);
        if (pa == a)
            return a;
        const int gp = lds_ld(&uf[pa]);
        if (gp != pa)
            lds_st(&uf[a], gp);
        a = pa;
    }
}

__device__ __forceinline__ void lds_union(int* uf, unsigned long long* c_fin, int a, int b)
{
    while (true)
    {
        a = lds_find(uf, a);
        b = lds_find(uf, b);
        if (a == b)
            return;
        if (a < b)
        {
            const int t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(&uf[a], a, b) == a)
        {
            atomicMax(&c_fin[b], lds_ld(&c_fin[a]));
            return;
        }
    }
}

// true iff some cluster's (lower-bounded) max finished_at has been passed by the column's minimum azimuth: only then can the
// finished-cluster check of cc.cpp:884-885 let a cluster through
__device__ __forceinline__ bool cluster_may_finish(LdsTrees& T, int n_unf, double min_az, double& lower_bound)
{
    bool may = false;
    double lb = 1.7976931348623157e308;
    for (int i = lane_id(); i < n_unf; i += 64)
        if (lds_ld(&T.uf[i]) == i)
        {
            const double f = __longlong_as_double((long long) lds_ld(&T.c_fin[i]));
            may |= !(f > min_az);
            lb = f < lb ? f : lb;
        }
    lower_bound = uniform_f64(wave_min_f64(lb)); // min over the clusters of (a lower bound of) their max finished_at
    return __any(may);
}

// exact single-lane replay of one column (rare): reference semantics with immediate attach / link, LDS tree state
template<int RPL>
__device__ void assoc_column_live(const AssocCtx& c, const cc_config& cfg, const Geometry& g, LdsTrees& T, int* s_win, const int lc,
                                  const long long gc, const int first_local, int& n_unf, double& L, long long& M, int& err, StreamState* st)
{
    const SP& p = c.p;
    const int R = c.R, RC = c.RC;
    int* wcol = s_win + (int) (gc % WIN_COLS) * R;
    for (int row = 0; row < R; row++)
        wcol[row] = -1;
    const CazBase cb = caz_base_of_column(gc, c.NC);
    for (int row = 0; row < R; row++)
    {
        const int pi = lc * R + row;
        if (p.ignored[pi])
        {
            p.root[pi] = -1;
            p.sc_parent[pi] = -2;
            if (g.mirror_fields)
                p.sc_visits[pi] = 0;
            continue;
        }
        const float mad = ccm::asinf_exact(cfg.max_distance / p.dist[pi]);
        const double pcaz = cell_caz(cb, p.incaz[pi]);
        const float4 me = p.sc_rec[pi];
        const float pincl = me.w, px = me.x, py = me.y, pz = me.z;
        int needed = f2i_x86(__builtin_ceilf(mad / c.az_width));
        needed = needed < c.max_steps_in_row ? needed : c.max_steps_in_row;
        int oc = lc;
        long long ogc = gc;
        int visits = 0, parcode = -1; // Point::number_of_visited_neighbors; the candidate whose child list the point joins (cc.cpp:663)
        int pslot = -1; // tree slot of the point (-1: none yet)
        for (int sb = 0; sb <= needed; sb++)
        {
            for (int dir = -1; dir <= 1; dir += 2)
            {
                if (dir == 1 && sb == 0)
                    continue;
                int sv = (dir == 1 || sb == 0) ? 1 : 0;
                int orow = (dir == 1 || sb == 0) ? row + dir : row;
                while (orow >= 0 && orow < R && sv <= c.max_steps_in_column)
                {
                    const int oi = oc * R + orow;
                    visits++; // cc.cpp:725
                    const float4 orec = p.sc_rec[oi];
                    if (ccm::absf(orec.w - pincl) > mad)
                        break;
                    if (!p.ignored[oi])
                    {
                        const int oslot = s_win[(int) (ogc % WIN_COLS) * R + orow]; // -2: finished tree
                        // cc.cpp:733: same root -> skip, unless the point's root sits in local column 0 (reference quirk; a
                        // same-tree candidate then only produces a self link, which is a no-op here)
                        const bool same = pslot >= 0 && oslot == pslot;
                        if (!same)
                        {
                            const float dx = px - orec.x, dy = py - orec.y, dz = pz - orec.z;
                            if (dx * dx + dy * dy + dz * dz < c.maxd2)
                            {
                                if (pslot == -1)
                                {
                                    if (oslot >= 0)
                                    {
                                        const uint32_t nw = (uint32_t) (gc - T.gcol[oslot] + 1);
                                        if (nw <= (uint32_t) c.NC)
                                        {
                                            pslot = oslot;
                                            parcode = (sb << 8) | orow;
                                            T.last[oslot] = (unsigned) gc;
                                            const unsigned long long cand = (unsigned long long) __double_as_longlong(pcaz + (double) mad);
                                            if (cand > T.fin[oslot])
                                                T.fin[oslot] = cand;
                                            atomicMax(&T.c_fin[lds_find(T.uf, oslot)], cand);
                                            T.pts[oslot]++;
                                        }
                                    }
                                }
                                else if (oslot >= 0 && oslot != pslot)
                                {
                                    log_link(g, st, p.link_log, T.cell[pslot], T.cell[oslot]);
                                    lds_union(T.uf, T.c_fin, pslot, oslot);
                                }
                            }
                        }
                    }
                    if (pslot != -1 && c.stop_enabled && sv >= c.stop_min_steps)
                        break;
                    orow += dir;
                    sv++;
                }
            }
            if (pslot != -1 && c.stop_enabled && sb >= c.stop_min_steps)
                break;
            if (oc == first_local)
                break;
            oc--;
            ogc--;
            if (oc < 0)
                oc += RC;
        }
        int rootcell;
        if (pslot == -1)
        {
            if (n_unf + 1 > TREE_SLOTS)
            {
                err = CC_ERR_CAPACITY;
                return;
            }
            pslot = n_unf;
            const double fin = pcaz + (double) mad;
            T.cell[pslot] = pi;
            T.gcol[pslot] = gc;
            T.fin[pslot] = (unsigned long long) __double_as_longlong(fin);
            T.last[pslot] = (unsigned) gc;
            T.pts[pslot] = 1;
            T.uf[pslot] = pslot;
            T.c_fin[pslot] = T.fin[pslot];
            if (n_unf == 0)
                M = gc;
            n_unf++;
            L = fin < L ? fin : L;
        }
        rootcell = T.cell[pslot];
        wcol[row] = pslot;
        p.root[pi] = rootcell;
        p.sc_parent[pi] = (int16_t) parcode; // the live scan's parent replaces the static one
        if (g.mirror_fields)
            p.sc_visits[pi] = sat_u16(visits);
    }
}

template<int RPL>
__global__ __launch_bounds__(64) void k_assoc_lds(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot)
{
    const int s = first_stream + blockIdx.x;
    const int lane = lane_id();
    StreamState* st = &states[s];
    if (st->error != 0 || st->batch[slot].seg_begin < 0 || st->assoc_mode != 0 || st->batch[slot].mode != 0 ||
        st->batch[slot].acp_next >= st->batch[slot].seg_end)
        return;
    AssocCtx c;
    c.p = stream_ptrs(P, g, s);
    const SP& p = c.p;
    const int R = c.R = g.num_rows;
    const int NC = c.NC = g.num_columns;
    const int RC = c.RC = g.ring_cols;
    c.az_width = g.az_width;
    c.maxd2 = g.max_distance_squared;
    c.max_steps_in_row = cfg.max_steps_in_row;
    c.max_steps_in_column = cfg.max_steps_in_column;
    c.stop_enabled = cfg.stop_after_association_enabled;
    c.stop_min_steps = cfg.stop_after_association_min_steps;
    const int nth = cfg.cluster_point_trees_every_nth_column;

    __shared__ LdsTrees T;
    __shared__ int s_win[WIN_COLS * WAVE * RPL];
    __shared__ int s_parent[WAVE * RPL];
    __shared__ int s_newslot[WAVE * RPL];
    __shared__ int s_bi[4];
    __shared__ double s_bd[2];
    __shared__ long long s_bl[2];

    long long first_unpub = st->first_unpublished, ring_start = st->ring_start;
    if (lane == 0 && st->batch[slot].pub_begin < 0)
        st->batch[slot].pub_begin = first_unpub; // first association kernel of this pass
    unsigned long long cluster_counter = st->cluster_counter;
    int n_unf = st->n_unfinished;
    long long M = st->min_required;
    double L = st->finish_lower_bound;
    double last_min_az = st->last_round_min_az;
    unsigned long long cells_published = st->cells_published, clusters_finished = st->clusters_finished;
    unsigned long long exceed = st->exceed_one_rotation, serial_cols = st->serial_columns, alias_rounds = st->stamp_alias_rounds;
    int n_events = st->n_events;
    const long long col_begin = st->batch[slot].acp_next, col_end = st->batch[slot].seg_end, first_column = st->first_column;
    int err = 0;
    long long err_a = 0, err_b = 0;
    const int tree_limit = g.lds_tree_limit;
    bool to_global = n_unf > tree_limit;
    __builtin_amdgcn_s_setprio(3); // latency-critical serial chain

    auto emit = [&](int type, long long a, long long b, unsigned cc, unsigned dd, long long column)
    {
        if (!g.record_events)
            return;
        if (lane == 0 && n_events < g.event_capacity)
        {
            cc_event e;
            e.type = type;
            e.stream = s;
            e.a = a;
            e.b = b;
            e.c = cc;
            e.d = dd;
            e.column = column;
            p.events[n_events] = e;
        }
        n_events++;
    };

    // ---- load the persistent tree state (global planes indexed by root cell) into LDS slots --------------------------------
    if (!to_global)
    {
        for (int i = lane; i < n_unf; i += 64)
        {
            const int cell = p.ulist[i];
            T.cell[i] = cell;
            const long long tg = p.colg[cell / R];
            T.gcol[i] = tg;
            T.fin[i] = (unsigned long long) __double_as_longlong(p.t_fin[cell]);
            T.last[i] = (unsigned) tg + p.t_width[cell] - 1u;
            T.pts[i] = p.t_pts[cell];
            T.uf[i] = p.t_pos[p.t_uf[cell]];
            T.c_fin[i] = T.fin[i];
        }
        wave_lds_fence();
        for (int i = lane; i < n_unf; i += 64)
            atomicMax(&T.c_fin[lds_find(T.uf, i)], T.fin[i]);
        // window of tree-slot ids for the WIN_COLS columns before col_begin: two dependent gathers per cell (root plane, then the
        // tree planes at the root) — issued 8 cells at a time so that a launch pays a few memory round trips, not one per cell
        constexpr int B = 8;
        for (int i0 = lane; i0 < WIN_COLS * R; i0 += 64 * B)
        {
            int rr[B];
#pragma unroll
            for (int u = 0; u < B; u++)
            {
                const int i = i0 + u * 64;
                rr[u] = -1;
                if (i < WIN_COLS * R)
                {
                    const int wc = i / R, row = i - wc * R;
                    // the global column in [col_begin - WIN_COLS, col_begin) that maps to window column wc
                    const long long gcx = col_begin - 1 - (((col_begin - 1) % WIN_COLS - wc + WIN_COLS) % WIN_COLS);
                    if (gcx >= first_column && gcx >= 0 && first_column >= 0)
                        rr[u] = p.root[(int) (gcx % RC) * R + row];
                }
            }
            int fin_[B], pos_[B];
#pragma unroll
            for (int u = 0; u < B; u++)
            {
                fin_[u] = 0;
                pos_[u] = -1;
                if (rr[u] >= 0)
                {
                    fin_[u] = p.t_finished[rr[u]];
                    pos_[u] = p.t_pos[rr[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < B; u++)
            {
                const int i = i0 + u * 64;
                if (i < WIN_COLS * R)
                    s_win[i] = rr[u] < 0 ? -1 : (fin_[u] ? -2 : pos_[u]);
            }
        }
    }
    __syncthreads();

    // staging of the next column (software prefetch; one global round trip per column stays off the critical path)
    int nx_parent[RPL], nx_nl[RPL];
    double nx_fin[RPL];
    unsigned long long nx_link[RPL];
    double nx_minaz = 0.;
    auto load_column = [&](long long gcx, int lcx)
    {
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            nx_parent[k] = -2;
            nx_nl[k] = 0;
            nx_fin[k] = 0.;
            nx_link[k] = 0;
            if (row < R && gcx < col_end)
            {
                const int ci = lcx * R + row;
                nx_parent[k] = p.sc_parent[ci];
                nx_nl[k] = p.sc_nlinks[ci];
                nx_fin[k] = p.sc_fin[ci];
                nx_link[k] = p.sc_links[ci];
            }
        }
        // lane 0 only: a divergent (vector) load. A uniform load would be a scalar SMEM load, whose latency every
        // later s_waitcnt lgkmcnt(0) (all LDS traffic) would have to sit out.
        if (lane == 0 && gcx < col_end)
            nx_minaz = p.colminaz[lcx];
    };
    int lc = (int) (col_begin % RC);
    int wcur = (int) (col_begin % WIN_COLS);
    int nth_phase = (int) (col_begin % nth);
    long long first_local_of = first_unpub;
    int first_local = (int) (first_unpub % RC);
    if (!to_global)
        load_column(col_begin, lc);

#ifdef CC_PROFILE_SECTIONS
    unsigned long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tmark = __builtin_amdgcn_s_memtime();
#define CC_SEC(i)                                                   \
    {                                                               \
        const unsigned long long _n = __builtin_amdgcn_s_memtime(); \
        tsec[i] += _n - tmark;                                      \
        tmark = _n;                                                 \
    }
#else
#define CC_SEC(i)
#endif
    long long gc = col_begin;
    CC_SEC(0)
    for (; gc < col_end && err == 0 && !to_global;
         gc++, lc = (lc + 1 == RC ? 0 : lc + 1), wcur = (wcur + 1) & (WIN_COLS - 1), nth_phase = (nth_phase + 1 == nth ? 0 : nth_phase + 1))
    {
        if (first_local_of != first_unpub)
        {
            const long long d = first_unpub - first_local_of;
            if (d > 0 && d < RC)
            {
                first_local += (int) d;
                if (first_local >= RC)
                    first_local -= RC;
            }
            else
                first_local = (int) (first_unpub % RC);
            first_local_of = first_unpub;
        }
        int parent[RPL], nl[RPL];
        unsigned long long link[RPL];
        double finc[RPL];
        const double min_az = uniform_f64(nx_minaz); // readfirstlane: lane 0 holds it, all lanes are active here
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            parent[k] = nx_parent[k];
            nl[k] = nx_nl[k];
            finc[k] = nx_fin[k];
            link[k] = nx_link[k];
        }
        CC_SEC(1)
        load_column(gc + 1, lc + 1 == RC ? 0 : lc + 1); // prefetch: nothing below depends on it
        CC_SEC(2)

        // ------------------------------------------------------------------ association (cc.cpp:773-835)
        bool bad = false; // any reason the static scan result may differ from the live scan for this column
        int cnt_new = 0;
        int newpos[RPL];
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            const bool is_new = parent[k] == -1;
            const unsigned long long mask = __ballot(is_new);
            newpos[k] = n_unf + cnt_new + __popcll(mask & lanes_below());
            cnt_new += __popcll(mask);
            if (row < R && RPL > 1)
            {
                // s_parent: row of the same-column parent, or the row itself when the chain ends here
                const bool same_col = parent[k] >= 0 && (parent[k] >> 8) == 0;
                s_parent[row] = same_col ? (parent[k] & 0xff) : row;
                s_newslot[row] = is_new ? newpos[k] : (parent[k] >= 0 ? -1 - parent[k] : 0x7fffffff);
            }
            if (nl[k] == 255)
                bad = true;
        }
        if (n_unf + cnt_new > tree_limit)
        {
            to_global = true; // continue this stream with the global-memory kernel, starting at this column
            break;
        }
        emit(CC_EV_GROUND_COLUMN, gc, gc, 0, 0, gc);
        if (RPL > 1)
            wave_lds_fence();
        CC_SEC(7)
        // pointer jumping: after ceil(log2(R)) rounds every row knows the top row of its same-column parent chain
        int top_of[RPL];
        if (RPL == 1)
        {
            // rows = lanes: jump through the cross-lane network (ds_bpermute), no LDS round trips
            const bool same_col = parent[0] >= 0 && (parent[0] >> 8) == 0;
            const int prow = parent[0] & 0xff;
            // Usual shape: the same-column parent of a row is the nearest non-ignored row above it. Then a chain is a run of
            // linked active rows and its top is the nearest active, unlinked row at or above — two ballots and a count of
            // leading zeros instead of pointer jumping through the cross-lane network.
            const unsigned long long active_m = __ballot(parent[0] >= -1);
            const unsigned long long linked_m = __ballot(same_col);
            const unsigned long long above = active_m & lanes_below();
            const int nearest_above = above ? 63 - __clzll((long long) above) : -1;
            if (!__any(same_col && prow != nearest_above))
            {
                const unsigned long long tops = active_m & ~linked_m & (lanes_below() | (1ull << lane));
                top_of[0] = tops ? 63 - __clzll((long long) tops) : lane;
            }
            else
            {
                int t = same_col ? prow : lane;
                for (int it = 0; it < 6; it++)
                {
                    const int t2 = __shfl(t, t);
                    const bool changed = t2 != t;
                    t = t2;
                    if (!__any(changed))
                        break;
                }
                top_of[0] = t;
            }
        }
        else
        {
#pragma unroll
            for (int it = 0; it < 7; it++)
            {
                int nxt[RPL];
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    nxt[k] = row < R ? s_parent[s_parent[row]] : 0;
                }
                wave_lds_fence();
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    if (row < R)
                        s_parent[row] = nxt[k];
                }
                wave_lds_fence();
            }
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                top_of[k] = row < R ? s_parent[row] : 0;
            }
        }
        int term_info = 0;
        if (RPL == 1)
        {
            const int mine = parent[0] == -1 ? newpos[0] : (parent[0] >= 0 ? -1 - parent[0] : 0x7fffffff);
            term_info = __shfl(mine, top_of[0]);
        }
        int slot[RPL];
        int freshcell[RPL]; // root cell of the point's tree
        // cc.cpp:657 (a tree may not span more than one rotation): M is the oldest start column of any unfinished tree, so while
        // gc - M + 1 <= NC no tree can fail the test and the per-lane look-up is skipped
        const bool span_check = n_unf > 0 && (uint32_t) (gc - M + 1) > (uint32_t) NC;
        // cc.cpp:762-763 (the live scan stops at the first unpublished column): only when the window reaches back that far
        const bool reach_check = gc - (WIN_COLS - 1) < first_unpub;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            slot[k] = -1;
            freshcell[k] = -1;
            if (parent[k] >= -1 && row < R)
            {
                const int top = top_of[k];
                // >= 0: new tree slot, < 0: -1 - code of a candidate in an earlier column
                const int tv = RPL == 1 ? term_info : s_newslot[top];
                int oldest_delta = 0;
                if (tv >= 0)
                {
                    slot[k] = tv;
                    freshcell[k] = lc * R + top;
                }
                else
                {
                    const int code = -1 - tv;
                    const int delta = code >> 8, prow = code & 0xff;
                    oldest_delta = delta;
                    const int v = s_win[((wcur - delta) & (WIN_COLS - 1)) * R + prow];
                    if (v < 0)
                        bad = true; // finished tree (attach refused, cc.cpp:658) or no tree
                    else
                    {
                        slot[k] = v;
                        freshcell[k] = T.cell[v];
                        if (span_check && (uint32_t) (gc - T.gcol[v] + 1) > (uint32_t) NC)
                            bad = true; // tree would span more than one rotation (cc.cpp:657)
                    }
                }
                // nothing may come from columns the live scan would not have reached (cc.cpp:762-763)
                if (reach_check)
                {
                    if (parent[k] >= 0)
                    {
                        const int pd = parent[k] >> 8;
                        oldest_delta = pd > oldest_delta ? pd : oldest_delta;
                        const int nlk = nl[k] == 255 ? 0 : nl[k];
#pragma unroll
                        for (int j = 0; j < LINK_SLOTS; j++)
                            if (j < nlk)
                            {
                                const int d = (int) ((link[k] >> (16 * j + 8)) & 0xff);
                                oldest_delta = d > oldest_delta ? d : oldest_delta;
                            }
                    }
                    if (gc - oldest_delta < first_unpub)
                        bad = true;
                }
            }
        }
        // (mirror mode) the static visit counts are only right if no scan looked past the first unpublished column
        if (g.mirror_fields && gc - ((p.col_info[lc] >> 24) & 0x7f) < first_unpub)
            bad = true;
        const bool column_live = __any(bad);
        CC_SEC(3)

        if (!column_live)
        {
            int* wcol = s_win + wcur * R;
            double l_new = L; // per-lane; L itself must stay wave-uniform (a divergent L drags the whole bookkeeping into VGPRs)
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (row < R)
                {
                    wcol[row] = slot[k];
                    // early: the store has a column of work to retire before the vmcnt(0) at the top of the next iteration
                    p.root[lc * R + row] = freshcell[k];
                    if (parent[k] == -1)
                    {
                        const int i = slot[k];
                        T.cell[i] = lc * R + row;
                        T.gcol[i] = gc;
                        T.fin[i] = (unsigned long long) __double_as_longlong(finc[k]);
                        T.last[i] = (unsigned) gc;
                        T.pts[i] = 1;
                        T.uf[i] = i;
                        T.c_fin[i] = (unsigned long long) __double_as_longlong(finc[k]);
                        l_new = finc[k] < l_new ? finc[k] : l_new;
                    }
                }
            }
            if (cnt_new > 0)
            {
                if (n_unf == 0)
                    M = gc;
                n_unf += cnt_new;
                L = uniform_f64(wave_min_f64(l_new));
            }
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                if (parent[k] >= 0)
                {
                    const int i = slot[k];
                    const int nlk = nl[k];
                    const int rep = lds_find(T.uf, i);
                    const unsigned long long fb = (unsigned long long) __double_as_longlong(finc[k]);
                    T.last[i] = (unsigned) gc;
                    atomicMax(&T.fin[i], fb);
                    atomicMax(&T.c_fin[rep], fb);
                    atomicAdd(&T.pts[i], 1u);
#pragma unroll
                    for (int j = 0; j < LINK_SLOTS; j++)
                        if (j < nlk)
                        {
                            const int code = (int) ((link[k] >> (16 * j)) & 0xffff);
                            const int v = s_win[((wcur - (code >> 8)) & (WIN_COLS - 1)) * R + (code & 0xff)];
                            if (v >= 0 && v != i)
                            {
                                log_link(g, st, p.link_log, T.cell[i], T.cell[v]);
                                lds_union(T.uf, T.c_fin, i, v);
                            }
                        }
                }
            }
            wave_lds_fence();
        }
        else
        {
            serial_cols++;
            if (lane == 0)
            {
                int nn = n_unf, e = 0;
                double LL = L;
                long long MM = M;
                assoc_column_live<RPL>(c, cfg, g, T, s_win, lc, gc, first_local, nn, LL, MM, e, st);
                s_bi[0] = nn;
                s_bi[1] = e;
                s_bd[0] = LL;
                s_bl[0] = MM;
            }
            wave_lds_fence();
            n_unf = uniform_i32(s_bi[0]);
            if (s_bi[1] == CC_ERR_CAPACITY)
            {
                // the live replay ran out of slots mid-column: this kernel cannot roll the column back
                err = CC_ERR_CAPACITY;
                err_a = n_unf;
            }
            L = uniform_f64(s_bd[0]);
            M = uniform_i64(s_bl[0]);
            wave_lds_fence();
        }
        if (err)
            break;

        CC_SEC(4)
        // ------------------------------------------------------------------ finished-cluster check (cc.cpp:837-974)
        if (nth_phase != 0)
            continue;
        CC_SEC(5)
        long long M_c;
        if (n_unf == 0)
            M_c = gc + 1;
        else if (min_az == last_min_az)
        {
            alias_rounds++;
            M_c = M;
        }
        else if (!((gc + 1 - M) >= NC) && (!(min_az >= L) || !cluster_may_finish(T, n_unf, min_az, L)))
            M_c = M; // nothing can be finished: first the scalar bound, then (refreshing it) the per-cluster bounds
        else
        {
            for (int i = lane; i < n_unf; i += 64)
            {
                T.a_fin[i] = 0ull;
                T.a_min[i] = 0x7fffffffffffffffll;
                T.a_max[i] = 0;
                T.a_pts[i] = 0;
                T.a_first[i] = 0x7fffffff;
                T.a_cid[i] = 0;
                T.a_flag[i] = 0;
            }
            wave_lds_fence();
            for (int i = lane; i < n_unf; i += 64)
            {
                const int j = lds_find(T.uf, i);
                T.comp[i] = j;
                atomicMax(&T.a_fin[j], T.fin[i]);
                atomicMin(&T.a_min[j], T.gcol[i]);
                atomicMax(&T.a_max[j], T.gcol[i] + (long long) (T.last[i] - (unsigned) T.gcol[i] + 1u));
                atomicAdd(&T.a_pts[j], T.pts[i]);
                atomicMin(&T.a_first[j], i);
            }
            wave_lds_fence();
            int exceed_local = 0, any_fin = 0;
            for (int i = lane; i < n_unf; i += 64)
                if (T.comp[i] == i)
                {
                    const double fin = __longlong_as_double((long long) T.a_fin[i]);
                    const bool unfinished = fin > min_az;
                    const bool exceeds = (T.a_max[i] - T.a_min[i]) >= NC;
                    if (exceeds)
                        exceed_local++;
                    const bool f = !unfinished || exceeds;
                    T.a_flag[i] = f ? 1 : 0;
                    any_fin |= f ? 1 : 0;
                }
            for (int o = 32; o > 0; o >>= 1)
                exceed_local += __shfl_xor(exceed_local, o);
            exceed += (unsigned long long) uniform_i32(exceed_local);
            wave_lds_fence();
            int last_first = -1;
            while (true)
            {
                int best = 0x7fffffff;
                for (int i = lane; i < n_unf; i += 64)
                    if (T.comp[i] == i && T.a_flag[i] && T.a_pts[i] > 5u)
                    {
                        const int fi = T.a_first[i];
                        if (fi > last_first && fi < best)
                            best = fi;
                    }
                best = uniform_i32(wave_min_i32(best));
                if (best == 0x7fffffff)
                    break;
                const int j = T.comp[best];
                const unsigned cid = (unsigned) cluster_counter;
                if (lane == 0)
                    T.a_cid[j] = cid;
                emit(CC_EV_CLUSTER, T.a_min[j], T.a_max[j] - 1, cid, T.a_pts[j], gc);
                cluster_counter++;
                clusters_finished++;
                last_first = best;
            }
            wave_lds_fence();
            // mark + persist finished trees, minimum required column, stable compaction of every slot array
            long long min_all = 0x7fffffffffffffffll, min_surv = 0x7fffffffffffffffll;
            double L_new = 1.7976931348623157e308;
            int out = 0;
            for (int base = 0; base < n_unf; base += 64)
            {
                const int i = base + lane;
                bool surv = false;
                int cell = 0, uf = 0;
                long long tg = 0;
                unsigned long long fin = 0, cfin = 0;
                unsigned width = 0, pts = 0;
                if (i < n_unf)
                {
                    cfin = T.a_fin[T.comp[i]]; // exact cluster maximum (only read at representatives)
                    cell = T.cell[i];
                    tg = T.gcol[i];
                    fin = T.fin[i];
                    width = T.last[i];
                    pts = T.pts[i];
                    uf = T.uf[i];
                    const int j = T.comp[i];
                    min_all = tg < min_all ? tg : min_all;
                    if (T.a_flag[j])
                    {
                        p.t_finished[cell] = 1;
                        p.t_cid[cell] = T.a_cid[j];
                        if (g.mirror_fields)
                        {
                            // final per-tree values of Point (cc.cpp:666-671) for the host mirror
                            p.t_fin[cell] = __longlong_as_double((long long) fin);
                            p.t_pts[cell] = pts;
                            p.t_width[cell] = (unsigned) (width - (unsigned) tg) + 1u;
                        }
                    }
                    else
                    {
                        surv = true;
                        min_surv = tg < min_surv ? tg : min_surv;
                        if (j == i)
                        {
                            const double f = __longlong_as_double((long long) T.a_fin[i]);
                            L_new = f < L_new ? f : L_new;
                        }
                    }
                }
                const unsigned long long mask = __ballot(surv);
                const int np = out + __popcll(mask & lanes_below());
                if (i < n_unf)
                    T.remap[i] = surv ? np : -2;
                wave_lds_fence();
                if (surv)
                {
                    T.cell[np] = cell;
                    T.gcol[np] = tg;
                    T.fin[np] = fin;
                    T.last[np] = width;
                    T.pts[np] = pts;
                    T.uf[np] = uf; // still an old position; remapped below
                    T.c_fin[np] = cfin;
                }
                out += __popcll(mask);
            }
            wave_lds_fence();
            out = uniform_i32(out);
            if (out != n_unf)
            {
                for (int i = lane; i < out; i += 64)
                    T.uf[i] = T.remap[T.uf[i]];
                for (int i = lane; i < WIN_COLS * R; i += 64)
                {
                    const int v = s_win[i];
                    if (v >= 0)
                        s_win[i] = T.remap[v];
                }
            }
            min_all = uniform_i64(wave_min_i64(min_all));
            min_surv = uniform_i64(wave_min_i64(min_surv));
            L = uniform_f64(wave_min_f64(L_new));
            M_c = min_all;
            M = min_surv;
            n_unf = out;
            wave_lds_fence();
        }
        last_min_az = min_az;

        // ------------------------------------------------------------------ publish bookkeeping (cc.cpp:1035-1092)
        if (M_c < first_unpub)
        {
            err = CC_ERR_BOOKKEEPING;
            err_a = M_c;
            err_b = first_unpub;
            break;
        }
        const long long old_unpub = first_unpub;
        first_unpub = M_c;
        ring_start = first_unpub - NC > 0 ? first_unpub - NC : 0;
        emit(CC_EV_PUBLISH_COLUMNS, old_unpub, first_unpub - 1, 0, 0, gc);
        cells_published += (unsigned long long) (first_unpub - old_unpub) * (unsigned long long) R;
        CC_SEC(6)
    }

    // ---- persist the tree state back to the global planes -------------------------------------------------------
    if (n_unf <= TREE_SLOTS)
    {
        wave_lds_fence();
        for (int i = lane; i < n_unf; i += 64)
        {
            const int cell = T.cell[i];
            p.ulist[i] = cell;
            p.t_pos[cell] = i;
            p.t_fin[cell] = __longlong_as_double((long long) T.fin[i]);
            p.t_width[cell] = T.last[i] - (unsigned) T.gcol[i] + 1u;
            p.t_pts[cell] = T.pts[i];
            p.t_uf[cell] = T.cell[T.uf[i]];
            p.t_cid[cell] = 0;
            p.t_finished[cell] = 0;
        }
    }
#ifdef CC_PROFILE_SECTIONS
    CC_SEC(7)
    if (lane == 0)
        for (int i = 0; i < 8; i++)
            st->dbg[8 + i] += tsec[i];
#endif
    if (lane == 0)
    {
        st->first_unpublished = first_unpub;
        st->batch[slot].pub_end = first_unpub;
        st->ring_start = ring_start;
        st->cluster_counter = cluster_counter;
        st->n_unfinished = n_unf;
        st->min_required = M;
        st->finish_lower_bound = L;
        st->last_round_min_az = last_min_az;
        st->cells_published = cells_published;
        st->clusters_finished = clusters_finished;
        st->exceed_one_rotation = exceed;
        st->serial_columns = serial_cols;
        st->stamp_alias_rounds = alias_rounds;
        st->batch[slot].acp_next = gc;
        st->n_events = n_events < g.event_capacity ? n_events : g.event_capacity;
        if (g.record_events && n_events > g.event_capacity && err == 0)
        {
            err = CC_ERR_CAPACITY;
            err_a = n_events;
        }
        if (to_global)
            st->assoc_mode = 1;
        if (err)
            raise_error(st, err, err_a, err_b);
    }
}

#include "cc_assoc_shared.h"
#include "cc_assoc3.h"
#include "cc_assocb.h"

// =====================================================================================================
// k_publish — cluster ids of the columns published in this pass: Point::id = id of the finished cluster of the point's
// tree (cc.cpp:1005). grid = (PUBLISH_BLOCKS, streams), block = 64, lanes = rows.
// =====================================================================================================
constexpr int PUBLISH_BLOCKS = 64;

// what a small call on the host path hands back (cc_engine.hip: add_firings_small), written straight into pinned host memory by the last kernel of
// the call instead of by three copy nodes of its graph: the stream's state, its first events, the early-stop counter
struct HostMirror
{
    StreamState* state;
    cc_event* events;
    int max_events;
    int* remaining;
    const int* d_remaining;
    unsigned long long* seq;   // pinned: the number of mirrored calls so far, written LAST (the host spins on it instead of synchronising the stream)
    unsigned long long* d_seq; // device: [0] that number, [1] blocks of the current launch that are through
};

// cluster ids of the columns the batch published (cc.cpp:1035-1092: what publishing leaves in Point::id), columns by .. ny .. strided
__device__ __forceinline__ void publish_body(const Geometry& g, const Planes& P, const StreamState* states, const int s, const int slot, const int by,
                                             const int ny)
{
    const StreamState* st = &states[s];
    if (st->batch[slot].pub_begin < 0)
        return;
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols;
    int plc = (int) ((st->batch[slot].pub_begin + by) % RC);
    const int plc_step = (int) ((unsigned) ny % (unsigned) RC);
    for (long long pc = st->batch[slot].pub_begin + by; pc < st->batch[slot].pub_end;
         pc += ny, plc = (plc + plc_step >= RC ? plc + plc_step - RC : plc + plc_step))
    {
        for (int row = lane_id(); row < R; row += 64)
        {
            const int ci = plc * R + row;
            const int r = p.root[ci];
            p.id[ci] = r >= 0 ? p.t_cid[r] : 0u;
        }
    }
}

// one wavefront: the call's results into pinned host memory, the sequence number last
__device__ __forceinline__ void mirror_results(const Geometry& g, const Planes& P, const StreamState* states, const int s, const HostMirror& hm)
{
    const int lane = lane_id();
    const StreamState* s0 = &states[s];
    const unsigned* src = (const unsigned*) s0;
    unsigned* dst = (unsigned*) hm.state;
    for (int i = lane; i < (int) (sizeof(StreamState) / 4); i += 64)
        dst[i] = src[i];
    const int ne = s0->n_events < hm.max_events ? s0->n_events : hm.max_events;
    const unsigned* es = (const unsigned*) (P.events + (size_t) s * g.event_capacity);
    unsigned* ed = (unsigned*) hm.events;
    for (int i = lane; i < ne * (int) (sizeof(cc_event) / 4); i += 64)
        ed[i] = es[i];
    if (lane == 0)
        *hm.remaining = *hm.d_remaining;
    __threadfence_system();
    if (lane == 0)
    {
        hm.d_seq[1] = 0ull;
        const unsigned long long v = hm.d_seq[0] + 1ull;
        hm.d_seq[0] = v;
        __hip_atomic_store(hm.seq, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(64) void k_publish(Geometry g, Planes P, const StreamState* states, int first_stream, int slot, HostMirror hm)
{
    publish_body(g, P, states, first_stream + (int) blockIdx.x, slot, (int) blockIdx.y, (int) gridDim.y);
    if (hm.state)
    {
        // the LAST block of the launch to get here mirrors the call's results: every cluster id of the call has been written by then, and the
        // association chain in front of this kernel left the state final
        __threadfence();
        unsigned long long through = 0;
        if (lane_id() == 0)
            through = atomicAdd(&hm.d_seq[1], 1ull);
        through = (unsigned long long) uniform_i64((long long) through);
        if (through == (unsigned long long) gridDim.x * gridDim.y - 1ull)
            mirror_results(g, P, states, first_stream, hm);
    }
}

// k_small_tail — what is behind the batch-parallel association in a call of a few firings on ONE stream (the per-column latency path): the exact serial
// kernel for whatever k_assocb left (nothing, normally), the streams that continue in global memory, the cluster ids of the published columns and
// the results into pinned host memory — k_assoc3 + k_publish in one launch (one graph node less: ~4.5 us of a 50 us call). grid = 1, block = A3_THREADS.
template<int RPL>
__global__ __launch_bounds__(A3_THREADS) void k_small_tail(Geometry g, cc_config cfg, Planes P, StreamState* states, int stream, int slot, HostMirror hm)
{
    assoc3_stream<RPL>(g, cfg, P, states, stream, slot, 0);
    __threadfence_block();
    __syncthreads(); // every wavefront has left the stream (its state is in the planes again)
    if (threadIdx.x < 64)
        associate_stream<RPL>(g, cfg, P, states, stream, slot);
    __syncthreads();
    publish_body(g, P, states, stream, slot, uniform_i32((int) (threadIdx.x >> 6)), (int) (blockDim.x >> 6));
    __syncthreads();
    if (hm.state && threadIdx.x < 64)
        mirror_results(g, P, states, stream, hm);
}


// =====================================================================================================
// k_scatter_info / k_scatter_apply — the frame scatter of the reference's harness (addColumnAndEvaluateFrameIfCompleted,
// kitti_demo.cpp:173-224) for a replayed KITTI sequence, on the device. A stream that is fed exactly num_columns pseudo-firings per frame
// (kitti_demo.cpp:386-403) carries, per cell, the sequence number of the firing that filled it: frame = sequence / num_columns, range-image
// column of the frame = sequence % num_columns, and the KITTI point of the cell is original_index[frame % slots][column][row]
// (cc_kitti_frame::d_original_index of the frame's conversion). k_scatter_info gives the smallest / largest frame among the points of every
// published column (what the harness needs to find where frame N + 1 starts, :205-209, and its two error conditions); k_scatter_apply
// writes is_ground_point = (ground_point_label == GP_GROUND) and detection_label = id (:214-215) of the columns' points into the frames'
// arrays in HBM, which cc_eval_frame_device then reads. grid = columns, block = 64 (lanes = rows).
// =====================================================================================================
__global__ __launch_bounds__(64) void k_scatter_info(Geometry g, Planes P, const StreamState* __restrict__ states, int s, long long from,
                                                     const int* __restrict__ original_index, int slots, int* __restrict__ out_min,
                                                     int* __restrict__ out_max)
{
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols, NC = g.num_columns;
    const long long gc = from + blockIdx.x;
    const int lc = (int) (gc % RC);
    const int* org = original_index + (size_t) s * (size_t) slots * (size_t) NC * (size_t) R;
    int mn = 0x7fffffff, mx = -1;
    // only published columns that are still in the ring hold what this reads (anything else: "no point", like an empty column)
    // (clearing is deferred by one call, so what a call published stays readable behind ring_start: the lower end is what has been CLEARED)
    const bool live = gc >= 0 && gc >= states[s].clear_done && gc < states[s].first_unpublished;
    for (int row = lane_id(); live && row < R; row += 64)
    {
        const int ci = lc * R + row;
        if (p.dist[ci] == p.dist[ci]) // the cell holds a return
        {
            const unsigned seq = p.src[ci];
            const int frame = (int) (seq / (unsigned) NC), col = (int) (seq % (unsigned) NC);
            if (org[((size_t) (frame % slots) * NC + col) * R + row] >= 0)
            {
                mn = frame < mn ? frame : mn;
                mx = frame > mx ? frame : mx;
            }
        }
    }
    mn = wave_min_i32(mn);
    mx = -wave_min_i32(-mx);
    if (lane_id() == 0)
    {
        out_min[blockIdx.x] = mn;
        out_max[blockIdx.x] = mx;
    }
}

__global__ __launch_bounds__(64) void k_scatter_apply(Geometry g, Planes P, const StreamState* __restrict__ states, int s, long long from,
                                                      const int* __restrict__ original_index, int slots, unsigned char* __restrict__ is_ground,
                                                      unsigned* __restrict__ detection, long long max_points)
{
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols, NC = g.num_columns;
    const long long gc = from + blockIdx.x;
    if (gc < 0 || gc < states[s].clear_done || gc >= states[s].first_unpublished)
        return; // (not a published column of the live ring)
    const int lc = (int) (gc % RC);
    const int* org = original_index + (size_t) s * (size_t) slots * (size_t) NC * (size_t) R;
    unsigned char* gr = is_ground + (size_t) s * (size_t) slots * (size_t) max_points;
    unsigned* det = detection + (size_t) s * (size_t) slots * (size_t) max_points;
    for (int row = lane_id(); row < R; row += 64)
    {
        const int ci = lc * R + row;
        if (p.dist[ci] == p.dist[ci])
        {
            const unsigned seq = p.src[ci];
            const int frame = (int) (seq / (unsigned) NC), col = (int) (seq % (unsigned) NC);
            const int pt = org[((size_t) (frame % slots) * NC + col) * R + row];
            if (pt >= 0 && pt < max_points)
            {
                const size_t o = (size_t) (frame % slots) * (size_t) max_points + (size_t) pt;
                gr[o] = p.ground[ci] == CC_GP_GROUND ? 1 : 0;
                det[o] = p.id[ci];
            }
        }
    }
}

// =====================================================================================================
// k_gather_clusters — member points of finished clusters, compacted on the device (the point gathering of
// collectPointsForCusterAndPublish, cc.cpp:985-1033): cluster i owns out[offset[i] .. offset[i] + n_points[i]) and receives its
// points in (global column, row) order. grid = clusters, block = 64 (lanes = rows), one pass over the cluster's column range.
// A point belongs to cluster c iff the root of its point tree carries c (t_cid, set when the cluster is finished).
// =====================================================================================================
struct ClusterQuery
{
    const unsigned* cid;       // [n] cluster ids (CC_EV_CLUSTER.c)
    const long long* col_from; // [n] first column (CC_EV_CLUSTER.a)
    const long long* col_to;   // [n] last column (CC_EV_CLUSTER.b)
    const long long* offset;   // [n] first output element of the cluster
    const unsigned* n_points;  // [n] expected number of points (CC_EV_CLUSTER.d)
    long long* out_gcol;
    int* out_row;
    int* mismatch; // incremented per cluster whose point count differs from n_points (columns cleared already, wrong descriptor)
};

__global__ __launch_bounds__(64) void k_gather_clusters(Geometry g, Planes P, const StreamState* states, int s, ClusterQuery q)
{
    const int ci_ = blockIdx.x;
    const SP p = stream_ptrs(P, g, s);
    const StreamState* st = &states[s];
    const int R = g.num_rows, RC = g.ring_cols;
    const int lane = lane_id();
    const unsigned cid = q.cid[ci_];
    const long long a = q.col_from[ci_], b = q.col_to[ci_];
    long long pos = q.offset[ci_];
    const long long end = pos + q.n_points[ci_];
    const bool readable = cid != 0 && a >= 0 && b >= a && b - a < RC && a >= st->clear_done && b <= st->ring_end;
    if (readable)
    {
        int lc = (int) (a % RC);
        for (long long gc = a; gc <= b; gc++, lc = (lc + 1 == RC ? 0 : lc + 1))
            for (int r0 = 0; r0 < R; r0 += 64)
            {
                const int row = r0 + lane;
                bool mine = false;
                if (row < R)
                {
                    const int cell = lc * R + row;
                    const int root = p.root[cell];
                    mine = p.colg[lc] == gc && root >= 0 && p.t_cid[root] == cid && p.t_finished[root];
                }
                const unsigned long long mask = __ballot(mine);
                if (mine)
                {
                    const long long o = pos + __popcll(mask & lanes_below());
                    if (o < end)
                    {
                        q.out_gcol[o] = gc;
                        q.out_row[o] = row;
                    }
                }
                pos += __popcll(mask);
            }
    }
    if (lane == 0 && pos != end)
        atomicAdd(q.mismatch, 1);
}

// =====================================================================================================
// k_view — host view of columns [from, from + ncols) of one stream (cc_engine_read_columns)
// grid = ncols, block = 64
// =====================================================================================================
struct ViewOut
{
    float *x, *y, *z, *dist, *incl;
    double* caz;
    int64_t *gcol, *src, *root_gcol;
    uint8_t *ground, *debug, *ignored;
    uint64_t* id;
    int32_t* root_row;
    // the remaining clustering fields of Point (include/cc_hip.h), any of them may be null
    double* fin;
    uint32_t *tpts, *width, *nchild;
    int32_t *visits, *par_row;
    uint8_t* finished;
    int64_t* par_gcol;
};

__global__ __launch_bounds__(64) void k_view(Geometry g, Planes P, const StreamState* states, int s, long long from, ViewOut o, int max_back)
{
    const StreamState* st = &states[s];
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols;
    const long long gc = from + blockIdx.x;
    const int lc = (int) (((gc % RC) + RC) % RC);
    const bool in_ring = st->ring_end >= 0 && gc >= 0 && gc >= st->clear_done && gc <= st->ring_end;
    const bool segmented = in_ring && st->first_column >= 0 && gc >= st->first_column && gc < st->first_unfinished;
    const CazBase cb = caz_base_of_column(gc >= 0 ? gc : 0, g.num_columns);
    const uint16_t tag = cell_tag((gc >= 0 ? gc : 0) / RC);
    for (int row = lane_id(); row < R; row += 64)
    {
        const size_t ci = (size_t) lc * R + row;
        const size_t oi = (size_t) blockIdx.x * R + row;
        const float nanf_ = __builtin_nanf("");
        const bool mine = p.gtag[ci] == tag; // the cell belongs to this pass over the ring (Point::global_column_index == gc)
        const bool filled = in_ring && (segmented ? true : mine);
        const bool has_point = filled && !(p.dist[ci] != p.dist[ci]) && mine;
        const float4 rec = has_point ? p.sc_rec[ci] : make_float4(nanf_, nanf_, nanf_, nanf_);
        o.x[oi] = rec.x;
        o.y[oi] = rec.y;
        o.z[oi] = rec.z;
        o.dist[oi] = has_point ? p.dist[ci] : nanf_;
        o.incl[oi] = (has_point || segmented) ? p.incl[ci] : nanf_;
        // (a segmented cell without a return sits in the middle of its column, cc.cpp:371-372)
        o.caz[oi] = has_point ? cell_caz(cb, p.incaz[ci]) : (segmented ? empty_cell_caz(gc, g.az_width) : __builtin_nan(""));
        o.gcol[oi] = segmented ? gc : (has_point ? gc : -1);
        // (the firing's sequence number, kept as its low 32 bits: it is one of the last 2^32 firings the stream consumed)
        o.src[oi] = has_point ? (long long) (st->firings_consumed - (unsigned long long) (uint32_t) ((uint32_t) st->firings_consumed - p.src[ci])) : -1;
        o.ground[oi] = segmented ? p.ground[ci] : (uint8_t) CC_GP_UNKNOWN;
        o.debug[oi] = segmented ? p.debug[ci] : (uint8_t) CC_DBG_WHITE;
        o.ignored[oi] = segmented ? p.ignored[ci] : 0;
        const int r = segmented ? p.root[ci] : -1;
        o.id[oi] = r >= 0 ? (uint64_t) p.t_cid[r] : 0ull;
        o.root_gcol[oi] = r >= 0 ? p.colg[r / R] : -1;
        o.root_row[oi] = r >= 0 ? r % R : 0;
        // per-tree values live at the root cell (cc.cpp:666-671, 818-822, 933); everything else keeps its cleared value
        const bool is_root = r >= 0 && (size_t) r == ci;
        if (o.fin)
            o.fin[oi] = is_root ? p.t_fin[ci] : 0.;
        if (o.tpts)
            o.tpts[oi] = is_root ? p.t_pts[ci] : 0u;
        if (o.width)
            o.width[oi] = is_root ? p.t_width[ci] : 0u;
        if (o.finished)
            o.finished[oi] = is_root ? p.t_finished[ci] : (uint8_t) 0;
        if (o.visits)
            o.visits[oi] = (segmented && g.mirror_fields) ? (int32_t) p.sc_visits[ci] : 0;
        const int code = (segmented && r >= 0) ? (int) p.sc_parent[ci] : -1; // (columns back << 8) | row of the point whose child list holds this one
        if (o.par_gcol)
            o.par_gcol[oi] = code >= 0 ? gc - (code >> 8) : -1;
        if (o.par_row)
            o.par_row[oi] = code >= 0 ? (code & 0xff) : 0;
    }
    if (o.nchild)
    {
        // Point::child_points.size(): the points of this and the following columns whose parent is a cell of this column
        __shared__ unsigned s_cnt[WAVE * MAX_ROWS_PER_LANE];
        for (int row = lane_id(); row < R; row += 64)
            s_cnt[row] = 0;
        __syncthreads();
        if (segmented)
            for (int d = 0; d <= max_back; d++)
            {
                const long long gd = gc + d;
                if (gd >= st->first_unfinished)
                    break;
                int ld = lc + d;
                ld = ld >= RC ? ld - RC : ld;
                for (int row = lane_id(); row < R; row += 64)
                {
                    const size_t cj = (size_t) ld * R + row;
                    const int code = p.root[cj] >= 0 ? (int) p.sc_parent[cj] : -1;
                    if (code >= 0 && (code >> 8) == d)
                        atomicAdd(&s_cnt[code & 0xff], 1u);
                }
            }
        __syncthreads();
        for (int row = lane_id(); row < R; row += 64)
            o.nchild[(size_t) blockIdx.x * R + row] = s_cnt[row];
    }
}

} // namespace cck
