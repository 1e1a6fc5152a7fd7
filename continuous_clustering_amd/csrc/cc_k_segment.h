// cc_k_segment.h — ground-point segmentation (cc.cpp:294-624): k_table, k_ego, k_seg_pre, k_seg_scan, k_seg_small.
// (part of cc_kernels.h: included there, in order, inside namespace cck)
#pragma once

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
// register budget of k_seg_scan as wavefronts per SIMD it must leave room for (2: 150 VGPRs, no spills; 4: 128 VGPRs, 18 spilled to scratch)
#ifndef CC_SEGSCAN_MIN_WAVES_PER_SIMD
#define CC_SEGSCAN_MIN_WAVES_PER_SIMD 2
#endif
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
__global__ __launch_bounds__(64, CC_SEGSCAN_MIN_WAVES_PER_SIMD) void k_seg_scan(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot)
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
