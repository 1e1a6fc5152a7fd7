// cc_k_base.h — wave-level helpers (DPP reductions, LDS accessors, uniform values), per-stream plane pointers, error reporting.
// (part of cc_kernels.h: included there, in order, inside namespace cck)
#pragma once

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
// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it waits for every global load in flight (prefetches) and
// for every global store to be acknowledged. For hand-offs through LDS between the wavefronts of a block; global memory is NOT ordered by it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
    int32_t* sl_cols;
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
    p.sl_cols = P.sl_cols + lo;
    return p;
}

// A point's contribution to its tree's finished_at (cc.cpp:700-703: continuous azimuth angle + the angle max_distance spans at the point's
// distance; 0 for a cell the association ignores). The window scan computes it for the points it packs (pk_fin); the serial kernels, which run
// for what the batch-parallel kernel left, recompute it from the cell's own planes with this — the same expression on the same operands, so the
// same bits — instead of the scan writing a double per cell that is never read in steady state (8 of the 24 B per cell it wrote, round 6).
// Launches in which the serial kernels are expected to associate a real share (the batch-parallel kernel switched off or stopping lately:
// Geometry::scan_stores_fin) keep the stored form, Planes::sc_fin: recomputing costs the serial kernels 17 % on vegetation.
__device__ __forceinline__ double cell_fin(const cc_config& cfg, const SP& p, const int ci, const CazBase& cb)
{
    if (p.ignored[ci])
        return 0.;
    const float mad = ccm::asinf_exact(cfg.max_distance / p.dist[ci]);
    return cell_caz(cb, p.incaz[ci]) + (double) mad;
}
__device__ __forceinline__ double cell_fin_of(const Geometry& g, const cc_config& cfg, const SP& p, const int ci, const CazBase& cb)
{
    return g.scan_stores_fin ? p.sc_fin[ci] : cell_fin(cfg, p, ci, cb);
}
