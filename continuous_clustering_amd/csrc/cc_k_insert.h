// cc_k_insert.h — continuous range-image insertion (cc.cpp:105-292): k_prep, k_insert2 (serial), k_insert_par / k_insert_par_fin (block-parallel, fused with the per-cell segmentation), k_insert_multi (multi-column firings).
// (part of cc_kernels.h: included there, in order, inside namespace cck)
#pragma once

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
// the three synchronisation words behind the LDS ring of staged firings (v_ready, v_done, v_stop)
template<int RPL>
__device__ __forceinline__ long long* insert2_sync_words(const int R)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    return (long long*) (smem + (size_t) ins_win_cols(RPL) * R * 4 + (size_t) INS_RING * R * 8 * 4);
}
// EMBEDDED (k_small_front): the caller has set the synchronisation words and run the block barrier behind them itself — every thread of ITS block
// has to meet that barrier, and only two of its wavefronts come here
template<int RPL, bool NOWIN = false, bool EMBEDDED = false>
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
    if (!EMBEDDED)
    {
        if (threadIdx.x == 0)
        {
            lds_st(v_ready, cursor0);
            lds_st(v_done, cursor0);
            lds_st(v_stop, -1ll);
        }
        __syncthreads();
    }

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
                    p.sc_rec[ci] = make_float4(r_x[so], r_y[so], r_z[so], r_i[so]);
                    p.inten[ci] = (uint8_t) r_t[so];
                    p.src[ci] = (uint32_t) (seq0 + (f - cursor0));
                    p.incl[ci] = r_i[so];
                    // rotation of the return = prev_rot + rot_off (cc.cpp:184-186) = that of its column gc = gcv (+ 1 if moved on), or one less
                    p.incaz[ci] = pack_incaz(r_a[so], cir[k] + (int) (gc - gcv[k]) >= NC);
                    p.gtag[ci] = cell_tag(pass);
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

// The same over the wavefronts of a block (k_insert_par_fin: the kernel behind the block-parallel insertion is on the chain a step of few streams
// waits for, and one wavefront walking a rotation's 35 tiles eight loads at a time was 23 of its ~55 us — tools/fin_probe.py): wavefront w takes
// the tiles [w C, (w + 1) C), finds the last valid entry of its chunk per row, the chunks' carries meet in LDS, every wavefront then writes its
// tiles with the carry that enters its chunk. Same values in the same places as table_from_partials. Every thread of the block calls it.
constexpr int TP_MAXC = 12; // tiles per wavefront this form takes (more: the serial form)
template<int RPL>
__device__ __forceinline__ void table_from_partials_waves(const SP& p, const int R, const bool fused, const int touched, const int ntiles, const int lane,
                                                          const int wave, const int nwaves)
{
    __shared__ float s_last[4][RPL * 64];
    __shared__ unsigned char s_has[4][RPL * 64];
    const int C = (touched + nwaves - 1) / nwaves;
    if (nwaves > 4 || C > TP_MAXC) // (wave-uniform: the arguments are)
    {
        if (wave == (nwaves > 1 ? 1 : 0))
            table_from_partials<RPL>(p, R, fused, touched, ntiles, lane);
        return;
    }
    const int t_lo = wave * C, t_hi = (t_lo + C < touched) ? t_lo + C : touched;
    float carry[RPL];
    unsigned long long v[TP_MAXC][RPL];
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        const int row = k * 64 + lane;
        carry[k] = (fused && row < R) ? p.curtab[row] : 0.f; // (read by every wavefront in front of the barrier; rewritten behind it)
    }
#pragma unroll
    for (int u = 0; u < TP_MAXC; u++)
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            v[u][k] = (row < R && t_lo + u < t_hi) ? p.tab_acc[(size_t) (t_lo + u) * R + row] : 0ull;
        }
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        float last = 0.f;
        unsigned char has = 0;
#pragma unroll
        for (int u = 0; u < TP_MAXC; u++)
            if (fused && t_lo + u < t_hi && t_lo + u < ntiles && (v[u][k] >> 32))
            {
                last = __uint_as_float((unsigned) v[u][k]);
                has = 1;
            }
        s_last[wave][k * 64 + lane] = last;
        s_has[wave][k * 64 + lane] = has;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RPL; k++)
        for (int w = 0; w < wave; w++)
            if (s_has[w][k * 64 + lane])
                carry[k] = s_last[w][k * 64 + lane];
#pragma unroll
    for (int u = 0; u < TP_MAXC; u++)
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            if (row < R && t_lo + u < t_hi)
            {
                if (v[u][k])
                    p.tab_acc[(size_t) (t_lo + u) * R + row] = 0ull;
                if (fused && t_lo + u < ntiles)
                {
                    p.tabc[(size_t) (t_lo + u) * R + row] = carry[k];
                    if (v[u][k] >> 32)
                        carry[k] = __uint_as_float((unsigned) v[u][k]);
                }
            }
        }
    if (fused && wave == nwaves - 1) // (the last chunk's carry-out is the table behind the batch; empty chunks pass their carry-in through)
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

// what one block of k_insert_par does behind its phase D, for launches that dealt a stream's firings to several blocks: take back what lies behind
// the first offending firing, then the stream state, the batch descriptor and (fused segmentation) the table. NW wavefronts, every thread of them.
template<int RPL>
__device__ __forceinline__ void par_fin_body(const Geometry& g, const SP& p, StreamState* st, const long long n, const int slot, int* __restrict__ left_over,
                                             const int fuse_on, const int nwaves, int* s_fused)
{
    const int lane = lane_id(), wave = uniform_i32((int) (threadIdx.x >> 6)), tid = threadIdx.x;
    const int R = g.num_rows, RC = g.ring_cols;
#ifdef CC_FIN_STATS
    // phase clocks (tools/fin_probe.py): entry | state read, barrier | thread 0: the stream's state | wavefront 1: the table
    const unsigned long long fin_t0 = __builtin_amdgcn_s_memtime();
#endif
    const int upto = st->par_upto;
    if (upto <= 0)
    {
        if (tid == 0 && st->par_clear_done >= 0)
            st->clear_done = st->par_clear_done; // (also for a stream that was not steady: its blocks shared the clearing all the same)
        return; // (not steady, or nothing taken: block 0 of k_insert_par has counted the stream as left over)
    }
    const bool fuse = fuse_on != 0 && left_over != nullptr && st->has_robot_tf != 0;
    const int bad = st->par_bad;
    const int done = bad < upto ? bad : upto;
    const long long prev_rear0 = st->prev_rearmost, first_unf0 = st->first_unfinished, ring_end0 = st->ring_end;
    const int lc0 = (int) (prev_rear0 % RC);
    const long long seq0 = (long long) st->firings_consumed;
    if (done < upto)
        par_take_back<RPL>(p, R, RC, lc0, p.par_off[done], p.par_off[upto - 1], wave, nwaves, lane);
    const bool whole = done == (int) n && done > 0;
    // (what par_close_stream will decide, known to every thread: the table does not wait for the state update — the two are chains of memory round
    // trips of their own, on wavefront 1 and on thread 0; one behind the other they made this kernel 55 us of a 340 us step at 32 streams)
    const bool fused_pred = left_over != nullptr && whole && fuse && ld_agent(&st->error) == 0;
    const int off_upto = p.par_off[upto - 1], off_done = done > 0 ? p.par_off[done - 1] : 0;
    __syncthreads(); // (everybody has read the state thread 0 is about to replace)
#ifdef CC_FIN_STATS
    const unsigned long long fin_t1 = __builtin_amdgcn_s_memtime();
#endif
    if (tid == 0)
    {
        st->clear_done = st->par_clear_done;
#ifndef CC_A2_STATS
        st->dbg[6] += (unsigned long long) done;
        st->dbg[7] += 1;
#endif
        *s_fused = par_close_stream(st, slot, left_over, fuse, whole, done, n, prev_rear0, first_unf0, ring_end0, seq0, (long long) off_done);
#ifdef CC_FIN_STATS
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long fin_t2 = __builtin_amdgcn_s_memtime();
        st->dbg[8] += fin_t1 - fin_t0;
        st->dbg[9] += fin_t2 - fin_t1;
        st->dbg[12] += 1;
#endif
    }
    if (fuse) // (every thread: the wavefronts share the tiles)
    {
        table_from_partials_waves<RPL>(p, R, fused_pred, (off_upto >> 6) + 1, fused_pred ? ((off_done + 63) >> 6) : 0, lane, wave, nwaves);
#ifdef CC_FIN_STATS
        __builtin_amdgcn_s_waitcnt(0);
        if (tid == 64)
            st->dbg[10] += __builtin_amdgcn_s_memtime() - fin_t1;
#endif
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
// register budget of the kernel as "wavefronts per SIMD it must leave room for" (512 VGPRs per lane and SIMD): what decides which of the other
// chains' blocks fit next to an insertion block on a compute unit (profiles/r06_kernel_resources.txt)
#ifndef CC_IP_MIN_WAVES_PER_SIMD
#define CC_IP_MIN_WAVES_PER_SIMD 1
#endif

// (W wavefronts per block: IP_WAVES next to the other chains' kernels; twice as many when a launch has few streams and the GPU is otherwise empty)
template<int RPL, int W = IP_WAVES>
__global__ __launch_bounds__(64 * W, CC_IP_MIN_WAVES_PER_SIMD) void k_insert_par(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream,
                                                            const float* __restrict__ xyz, const uint8_t* __restrict__ inten,
                                                            const double* __restrict__ poses, long long n, long long n_total, long long fbase,
                                                            int slot, int* __restrict__ left_over, const double* __restrict__ ego,
                                                            const int* __restrict__ prev_left = nullptr)
{
    // prev_left (the engine's lazy gate, cc_engine.hip): the counters the PREVIOUS batch's insertion left behind. The host enqueues this batch's
    // insertion before it has read them; if they say that the previous batch needs the other insertion kernels (or k_table on this chain), this
    // launch must not have happened: it leaves without a trace and the host launches it again behind those kernels.
    if (prev_left && (prev_left[0] | prev_left[1]) != 0)
        return;
    const int sl = blockIdx.x;
    const int s = first_stream + sl;
    const int lane = lane_id();
    const int wave = uniform_i32((int) (threadIdx.x >> 6)); // (readfirstlane: the firing index and everything addressed with it stay scalar)
    const int tid = threadIdx.x;
    StreamState* st = &states[s];
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, NC = g.num_columns, RC = g.ring_cols;
#ifdef CC_IP_STATS
    // phase clocks of the block's wavefront 0 (tools/ip_probe.py): entry | clearing | 0 | B | D | end
    unsigned long long ip_t[6];
    ip_t[0] = __builtin_amdgcn_s_memtime();
#define IP_MARK(i) ip_t[i] = __builtin_amdgcn_s_memtime();
#else
#define IP_MARK(i)
#endif
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
    if (clear_done >= 0)
    {
        // (every block of the stream takes a share: the columns cleared here lie at or above clear_done_entry, the slots the run writes have their
        // previous tenant below it — no block's cells can meet another block's clearing. One block alone needed 27 us for a rotation's columns, which
        // the launch of 32 streams x 4 blocks then lasted longer than its other blocks)
        long long clear_to = ring_start < st->clear_allowed ? ring_start : st->clear_allowed;
        if (nby > 1)
        {
            // the blocks of a stream start at different times and the association chain of the previous batch may be advancing ring_start meanwhile:
            // the first block to arrive fixes the limit for all of them (StreamState::par_clear_done, -1 since k_begin_batch). With a limit of its
            // own a block would leave its share of [its view, block 0's view) uncleared below the clear_done block 0 reports.
            __shared__ long long s_clear_to;
            if (tid == 0)
            {
                const long long want = clear_to > clear_done ? clear_to : clear_done;
                const unsigned long long seen = atomicCAS((unsigned long long*) &st->par_clear_done, ~0ull, (unsigned long long) want);
                s_clear_to = seen == ~0ull ? want : (long long) seen;
            }
            __syncthreads();
            clear_to = s_clear_to;
        }
        const int cstep = W * nby;
        int clc = (int) ((clear_done + wave + W * by) % RC);
        for (long long c = clear_done + wave + W * by; c < clear_to; c += cstep, clc = clc + cstep >= RC ? clc + cstep - RC : clc + cstep)
        {
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
            // (several blocks: the others may not have read clear_done yet — a block that found the new value would skip its share of the
            // clearing; k_insert_par_fin takes it from par_clear_done)
            if (nby == 1)
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

    IP_MARK(1)
    // ---- 0: the column of every firing from its first valid return (prep_point's column arithmetic, nothing else of it)
    // (one lane per firing: the lanes' loads are 768 B apart, a round trip to L2 / HBM each — the first rows of four rounds are requested together)
    constexpr int IP_PF = 4;
    for (int f0 = tid; f0 < nn; f0 += 64 * W * IP_PF)
    {
        float hx[IP_PF], hy[IP_PF];
#pragma unroll
        for (int k = 0; k < IP_PF; k++)
        {
            const int f = f0 + k * 64 * W;
            hx[k] = hy[k] = __builtin_nanf("");
            if (f < nn)
            {
                const size_t base = (fglob + (size_t) f) * R * 3;
                hx[k] = xyz[base];
                hy[k] = xyz[base + 1];
            }
        }
#pragma unroll
        for (int k = 0; k < IP_PF; k++)
        {
            const int f = f0 + k * 64 * W;
            if (f >= nn)
                break;
            const size_t base = (fglob + (size_t) f) * R * 3;
            float fx = hx[k], fy = hy[k];
            bool found = fx == fx;
            for (int row = 1; row < R && !found; row++)
            {
                fx = xyz[base + (size_t) row * 3];
                if (fx == fx)
                {
                    fy = xyz[base + (size_t) row * 3 + 1];
                    found = true;
                }
            }
            int c = -1;
            if (found)
            {
                const float az = ccm::atan2f_exact(fy, fx);
                const float inc_az = clockwise ? -az + CC_PI_F : az + CC_PI_F;
                c = f2i_x86(inc_az / g.az_width);
            }
            s_c[f] = (short) ((c >= 0 && c < NC && c < 32768) ? c : -1);
        }
    }
    __syncthreads();
    IP_MARK(2)
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
    IP_MARK(3)
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
    IP_MARK(4)
    __syncthreads();
#ifdef CC_IP_STATS
    IP_MARK(5)
    if (tid == 0 && by == 0)
    {
        for (int i = 0; i < 5; i++)
            atomicAdd(&st->dbg[8 + i], ip_t[i + 1] - ip_t[i]);
        atomicAdd(&st->dbg[13], 1ull);
    }
#endif
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

// k_insert_par_fin — par_fin_body as a kernel: a kernel boundary is what makes the blocks' writes visible to it for free. (Round 6 also built the
// other way — the stream's last block through doing this work, the launch's last stream writing the host's gate counters: exact, but the agent-scope
// fences the hand-over needs write the XCD's whole L2 back, - 22 % at 32 streams, and the gate's arguments doubled the kernel's scalar-register
// spills; taken out again.) grid = streams, block = 256.
template<int RPL>
__global__ __launch_bounds__(256) void k_insert_par_fin(Geometry g, Planes P, StreamState* states, int first_stream, const float* __restrict__ xyz,
                                                        long long n, long long n_total, long long fbase, int slot, int* __restrict__ left_over, int fuse_on,
                                                        const int* __restrict__ prev_left = nullptr)
{
    if (prev_left && (prev_left[0] | prev_left[1]) != 0)
        return; // (see k_insert_par)
    (void) xyz;
    (void) n_total;
    (void) fbase;
    const int s = first_stream + (int) blockIdx.x;
    __shared__ int s_fused;
    par_fin_body<RPL>(g, stream_ptrs(P, g, s), &states[s], n, slot, left_over, fuse_on, 4, &s_fused);
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
// Open columns staged in LDS (round 5). A firing's returns land in up to ~60 different columns, every one of them 4 (or 1, 2) bytes into a 32-byte
// sector of its own that is evicted from L2 half empty long before the column's other rows arrive: seven planes x one sector per cell = 7.2 GB
// written per 256 x 1700-firing launch for 1.9 GB of cells. The planes named here are kept in a ring of IM_W column slots in LDS instead and go out
// when the rearmost laser has passed their column — whole columns, rows as lanes, every sector complete: distance (4 bytes), intensity (1), and one byte
// 0x80 | (firing & 127) from which the source firing (4 bytes) and the cell's ring tag (2) follow (a cell waits fewer than IM_W + IM_WAVES < 128
// firings). -DCC_IM_STAGE_INCAZ=1 also keeps the azimuth-in-column plane there (36 KB more LDS at 128 rows).
#ifndef CC_IM_W
#define CC_IM_W 72
#endif
#ifndef CC_IM_STAGE_INCAZ
#define CC_IM_STAGE_INCAZ 0
#endif
#ifndef CC_IM_STAGE_DIST
#define CC_IM_STAGE_DIST 1
#endif
#ifndef CC_IM_STAGE
#define CC_IM_STAGE 1 // (0: every plane straight to global memory, as until round 5 — for A/B builds, tools/build_variant.sh)
#endif
constexpr int IM_W = CC_IM_W; // >= the widest firing accepted + the columns the rearmost laser advances within a chunk + 1
static_assert(IM_W + IM_WAVES < 128, "the staged firing index has seven bits");

template<int RPL>
__global__ __launch_bounds__(64 * IM_WAVES) void k_insert_multi(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream,
                                                              const float* __restrict__ xyz, const uint8_t* __restrict__ inten,
                                                              const double* __restrict__ poses, long long n, long long n_total, long long fbase,
                                                              int slot, int* __restrict__ left_over)
{
    // left_over (engine option skip_idle_fallbacks, launches in which this is the first insertion kernel): a stream whose whole batch is taken here
    // gets its batch descriptor here (as in k_insert_par); every other stream is counted, and the host launches k_prep + k_insert2 only if there is one
#ifdef CC_IM_STATS
    const unsigned long long im_t0 = __builtin_amdgcn_s_memtime();
#endif
    const int sl = blockIdx.x;
    const int s = first_stream + sl;
    const int lane = lane_id();
    const int wave = uniform_i32((int) (threadIdx.x >> 6));
    const int tid = threadIdx.x;
    StreamState* st = &states[s];
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, NC = g.num_columns, RC = g.ring_cols;
    constexpr int NR = 64 * RPL;
    __shared__ int2 s_pair[IM_WAVES];    // x: column-in-rotation of the firing's rearmost laser, -1 = empty firing; y: foremost - rearmost column of the firing
    __shared__ int s_col[IM_WAVES][NR];  // per row: columns ahead of the firing's rearmost laser, -1 = no return
    __shared__ int s_rowmax[NR];         // per row: last column written (relative to prev_rearmost at entry), INT_MIN = none in reach
    __shared__ int s_stop;               // first firing of the chunk whose rows clash with earlier returns
    __shared__ long long s_ring_start;
#if CC_IM_STAGE
#if CC_IM_STAGE_DIST
    __shared__ float s_dist[IM_W][NR];         // the open columns: distance ...
#endif
    __shared__ unsigned char s_fw[IM_W][NR];   // ... 0 = not written by this launch, else 0x80 | (firing of the batch & 127)
    __shared__ unsigned char s_int[IM_W][NR];  // ... intensity
#if CC_IM_STAGE_INCAZ
    __shared__ float s_incaz[IM_W][NR];
#endif
#else
    __shared__ unsigned char s_fw[1][4];
#endif

    // (the stream's state is the same in every lane: through readfirstlane the chunk walks below stay on the scalar unit)
    const long long cursor0 = uniform_i64(st->cursor);
    if (cursor0 >= n)
    {
        if (left_over && tid == 0)
            atomicAdd(left_over, 1); // (an empty call, or a batch another kernel closed: the serial kernel writes the descriptor)
        return;
    }
    const long long prev_rear0 = uniform_i64(st->prev_rearmost), prev_fore0 = uniform_i64(st->prev_foremost), first_unf0 = uniform_i64(st->first_unfinished);
    const long long ring_end0 = uniform_i64(st->ring_end);
    if (tid == 0)
        s_ring_start = __hip_atomic_load(&st->ring_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const long long ring_start = uniform_i64(s_ring_start);
    // deferred clearColumns (cc.cpp:1094-1145), spread over the wavefronts (as in k_insert_par; nothing left to do when that kernel ran)
    long long clear_done = uniform_i64(st->clear_done);
    if (clear_done >= 0)
    {
        const long long clear_allowed = uniform_i64(st->clear_allowed);
        const long long clear_to = ring_start < clear_allowed ? ring_start : clear_allowed;
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
#if CC_IM_STAGE
    for (int i = tid; i < IM_W * NR / 4; i += 64 * IM_WAVES)
        ((unsigned*) &s_fw[0][0])[i] = 0u;
#endif
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
    const long long seq0 = uniform_i64((long long) st->firings_consumed);
    const long long rot0 = prev_rear0 / NC;
    const int cir0 = (int) (prev_rear0 - rot0 * NC);
    const int lc0 = (int) (prev_rear0 % RC);
    const long long pass0 = prev_rear0 / RC; // pass over the ring of the previous rearmost laser (cell_tag)
    // the two 64-bit limits of the walk below as columns relative to prev_rear0 (32 bits: the walk is ~50 scalar instructions per firing on the
    // chunk's critical path): the emission limit (k_insert2's loop head: (prev_rear0 + rel) - first_unf0 >= limit_columns ends the run) and the first
    // column whose ring slot is not known to be cleared (prev_rear0 + rel' - RC >= clear_done)
    auto clamp_rel = [](const long long v) { return v < -0x40000000ll ? -0x40000000ll : (v > 0x40000000ll ? 0x40000000ll : v); };
    const int lim_rel = (int) clamp_rel((long long) g.limit_columns - (prev_rear0 - first_unf0));
    const int clr_rel = (int) clamp_rel(clear_done + (long long) RC - prev_rear0);
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
    // the staged columns: slot base_slot holds column base_rel (relative to prev_rear0), everything behind it has been written out
    int base_rel = 0, base_slot = 0;
    // write out the staged columns [base_rel, upto) — one wavefront per column, rows as lanes — and free their slots; last_frel: the batch-relative
    // index of the last firing accepted so far (the staged seven bits are completed from it)
    auto flush_columns = [&](const int upto, const long long last_frel)
    {
#if CC_IM_STAGE
        for (int c = base_rel + wave; c < upto; c += IM_WAVES)
        {
            int slot = base_slot + (c - base_rel);
            slot = slot >= IM_W ? slot - IM_W : slot;
            const unsigned lcq = (unsigned) (lc0 + c) / (unsigned) RC;
            const int lc = (int) ((unsigned) (lc0 + c) - lcq * (unsigned) RC);
            const uint16_t tag = cell_tag(pass0 + (long long) lcq);
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                const unsigned fw = s_fw[slot][row];
                if (fw)
                {
                    const unsigned ci = (unsigned) lc * (unsigned) R + (unsigned) row;
                    const long long frel = last_frel - (long long) (((unsigned) last_frel - fw) & 127u);
#if CC_IM_STAGE_DIST
                    at32(p.dist, ci) = s_dist[slot][row];
#endif
                    at32(p.inten, ci) = s_int[slot][row];
                    at32(p.src, ci) = (uint32_t) (seq0 + frel);
                    at32(p.gtag, ci) = tag;
#if CC_IM_STAGE_INCAZ
                    at32(p.incaz, ci) = s_incaz[slot][row];
#endif
                    s_fw[slot][row] = 0;
                }
            }
        }
#endif
        if (upto > base_rel)
        {
            base_slot = (base_slot + (upto - base_rel)) % IM_W;
            base_rel = upto;
        }
    };
    load_firing(cursor0 + wave);
#ifdef CC_IM_STATS
    // phase clocks of one wavefront (tools/im_probe.py): prepare | wait 1 | walk + collision rule | wait 2 | cells | carry + wait 3
    unsigned long long im_acc[6] = {0, 0, 0, 0, 0, 0}, im_t = __builtin_amdgcn_s_memtime(), im_chunks = 0;
    const unsigned long long im_prologue = im_t - im_t0; // (entry, clearing, the rows' last columns)
#define IM_MARK(i)                                                \
    {                                                             \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        im_acc[i] += t_ - im_t;                                   \
        im_t = t_;                                                \
    }
#else
#define IM_MARK(i)
#endif
    for (long long f0 = cursor0; f0 < n; f0 += IM_WAVES)
    {
        const long long f = f0 + wave;
        const bool mine = f < n;
        // the columns the previous chunk's firings left behind them are complete (behind the barrier that ended that chunk; the two barriers of this
        // chunk lie between these reads and the next writes into the freed slots)
        flush_columns(carry_rel, f0 - cursor0 - 1);
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
        // (wave-uniform: the matrix travels in SGPRs, by scalar loads — as in k_insert_par. Requested one chunk ahead like the returns it made the
        // preparation 3.8 k clocks per chunk SLOWER: scalar loads return out of order, so the first wait for any other scalar load waits for it too)
        double T[12];
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
            s_pair[wave] = make_int2(rear_cir, span);
            if (wave == 0)
                s_stop = IM_WAVES;
        }
        IM_MARK(0)
        lds_barrier();
        IM_MARK(1)
        // Everything the chunk's hand-off arrays hold is requested at once: lane j reads firing j's (rear column, span) pair and the values travel to
        // scalar registers by v_readlane, the earlier firings' per-row columns and the row's running maximum sit in vector registers before the first of
        // them is used. (Round 5, tools/im_probe.py: read inside the walks — one dependent LDS round trip per firing and step — the rear-column walk
        // cost 4.9 k clocks per chunk and the collision rule of the chunk's last wavefront another 3.9 k, of 21 k.)
        int rcs[IM_WAVES], sps[IM_WAVES];
        {
            const int2 pr = s_pair[lane & (IM_WAVES - 1)];
#pragma unroll
            for (int j = 0; j < IM_WAVES; j++)
            {
                rcs[j] = __builtin_amdgcn_readlane(pr.x, j);
                sps[j] = __builtin_amdgcn_readlane(pr.y, j);
            }
        }
        int ocj[RPL][IM_WAVES - 1], rowmax_in[RPL];
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            rowmax_in[k] = s_rowmax[k * 64 + lane];
#pragma unroll
            for (int j = 0; j < IM_WAVES - 1; j++)
                ocj[k][j] = s_col[j][k * 64 + lane];
        }
        // ---- rear column of every firing of the chunk (every thread the same scalar walk), first firing that ends the run -------------
        int my_rel = 0, my_prev_rel = 0, stop = IM_WAVES;
        int relj[IM_WAVES]; // rear column of firing j relative to prev_rear0 (valid below `stop`)
        {
            int rel = carry_rel, cir = carry_cir;
            bool open = true;
            const int in_batch = n - f0 < IM_WAVES ? (int) (n - f0) : IM_WAVES;
#pragma unroll
            for (int j = 0; j < IM_WAVES; j++)
            {
                const int rc = rcs[j], sp = sps[j];
                const int diff = rc - cir;
                const bool ok = rc >= 0 && ((diff > 0 && diff <= half) || diff < -half) && sp < half; // strictly forward, also across the wrap
                const int delta = ok ? (diff < -half ? diff + NC : diff) : 0;
                const int nrel = rel + delta;
                // taken only while the batch has emitted fewer than limit_columns columns before it (k_insert2's loop head), and while the
                // previous tenant of every ring slot it touches is known to be cleared
                // ... and while all its cells lie inside the staged window (base_rel is the rear column the previous chunk ended at)
                const bool take = open && j < in_batch && ok && rel < lim_rel && nrel + sp < clr_rel && (!CC_IM_STAGE || nrel + sp - base_rel < IM_W);
                if (open && !take)
                {
                    stop = j;
                    open = false;
                }
                relj[j] = nrel;
                if (take && j == wave)
                {
                    my_rel = nrel;
                    my_prev_rel = rel;
                }
                if (take)
                {
                    rel = nrel;
                    cir = rc;
                }
            }
        }
        // ---- per-row collision rule: this firing's cell of a row must lie ahead of everything the row has written -----------------
        if (mine && wave < stop)
        {
            bool clash = false;
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                if (oc[k] >= 0)
                {
                    int last = rowmax_in[k];
#pragma unroll
                    for (int j = 0; j < IM_WAVES - 1; j++) // (the earlier firings of the chunk)
                    {
                        const int o = ocj[k][j];
                        if (j < wave && o >= 0)
                            last = relj[j] + o > last ? relj[j] + o : last;
                    }
                    clash |= my_rel + oc[k] <= last;
                }
            }
            if (__any(clash) && lane == 0)
                atomicMin(&s_stop, wave);
        }
        IM_MARK(2)
        lds_barrier();
        IM_MARK(3)
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
                    at32(p.incl, ci) = q[k].incl;
#if !CC_IM_STAGE
                    at32(p.inten, ci) = cint[k];
                    at32(p.src, ci) = (uint32_t) (seq0 + (f - cursor0));
                    at32(p.dist, ci) = q[k].dist;
                    at32(p.incaz, ci) = q[k].incaz;
                    at32(p.gtag, ci) = cell_tag(pass0 + (long long) lcq);
#else
                    // (distance, intensity, source firing and ring tag wait in LDS until the column is complete: flush_columns)
                    int slot = base_slot + (crel - base_rel);
                    slot = slot >= IM_W ? slot - IM_W : slot;
#if CC_IM_STAGE_DIST
                    s_dist[slot][row] = q[k].dist;
#else
                    at32(p.dist, ci) = q[k].dist;
#endif
                    s_int[slot][row] = cint[k];
                    s_fw[slot][row] = (unsigned char) (0x80u | ((unsigned) (f - cursor0) & 127u));
#if CC_IM_STAGE_INCAZ
                    s_incaz[slot][row] = q[k].incaz;
#else
                    at32(p.incaz, ci) = q[k].incaz; // (the return's rotation, rot0 + (cir0 + crel) / num_columns, is that of its column)
#endif
#endif
                    atomicMax(&s_rowmax[row], crel);
                }
            }
            // columns [G_(f-1), G_f) (rearmost columns) are finished by this firing and carry its pose (cc.cpp:289-291)
            const int cnt = my_rel - my_prev_rel;
            for (int jj = lane; jj < cnt; jj += 64)
                p.trig[(int) ((unsigned) (lc0 + my_prev_rel + jj) % (unsigned) RC)] = (int) f;
        }
        IM_MARK(4)
        // ---- carry (the same walk over the accepted firings, in every thread) -------------------------------------------------------
#pragma unroll
        for (int j = 0; j < IM_WAVES; j++)
            if (j < stop)
            {
                carry_rel = relj[j];
                carry_cir = rcs[j];
                fore_rel = relj[j] + sps[j] > fore_rel ? relj[j] + sps[j] : fore_rel;
            }
        done = f0 + stop;
        if (stop < IM_WAVES)
            break;
        lds_barrier(); // this chunk's LDS reads and s_rowmax updates are complete before the next chunk rewrites the hand-off arrays (LDS only:
                       // __syncthreads() would also wait for the cells' stores to be acknowledged and for the next firing's loads — 2 - 4 k clocks per chunk)
#ifdef CC_IM_STATS
        IM_MARK(5)
        im_chunks++;
#endif
    }
    __syncthreads();
    flush_columns(base_rel + IM_W, done - cursor0 - 1); // whatever is still staged: the open columns in front of the rearmost laser
#ifdef CC_IM_STATS
#ifndef CC_IM_STATS_WAVE
#define CC_IM_STATS_WAVE 0
#endif
    if (wave == CC_IM_STATS_WAVE && lane == 0)
    {
        for (int i = 0; i < 6; i++)
            atomicAdd(&st->dbg[8 + i], im_acc[i]);
        atomicAdd(&st->dbg[14], im_chunks);
        atomicAdd(&st->dbg[15], im_prologue);
        atomicAdd(&st->dbg[5], 1ull);
    }
#endif
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
