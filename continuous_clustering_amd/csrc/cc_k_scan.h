// cc_k_scan.h — the window scan as a pure function of static data: k_scan (rows as lanes), k_scan2 (active points packed into the lanes), k_small_front (front half of a small call in one launch).
// (part of cc_kernels.h: included there, in order, inside namespace cck)
#pragma once

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
                if (g.scan_stores_fin)
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
                    if (g.scan_stores_fin)
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
    if (threadIdx.x == 0)
    {
        // (what insert2_body does in front of its own barrier when it is a kernel of its own: the batch starts at firing 0)
        long long* w = insert2_sync_words<1>(R);
        lds_st(w, 0ll);
        lds_st(w + 1, 0ll);
        lds_st(w + 2, -1ll);
    }
    __syncthreads(); // (workgroup-scope release / acquire: the staging planes, the ego records and the stream state are visible to wavefronts 0 and 1)
    SF_MARK(1)
    if (threadIdx.x < 128)
        insert2_body<1, true, true>(g, cfg, P, states, stream, slot, inten, n, remaining, n, 0, 0);
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

// ---- the per-lane state machine of one point's window scan (cc.cpp:706-769), shared by k_scan2 and k_scan2_long ----------------------------------
// column offset sb, direction (0 = rows above, 1 = rows below), vertical step d; one visit per call of step()
struct ScanCfg
{
    int R, RC, max_col_steps, stop_min;
    bool stop_enabled;
    float maxd2;
};

struct ScanPoint
{
    float4 me;
    float mad;
    int row, needed, bound;
    int sb, down, d, oc, orow;
    int rooted, parent, nlinks, overflow, visits, reach;
    unsigned long long packed;
    bool run;

    __device__ __forceinline__ void next_column(const ScanCfg& c) // the end of a column's visits: cc.cpp:756-769
    {
        if ((rooted && c.stop_enabled && sb >= c.stop_min) || oc == bound || sb + 1 > needed)
            run = false;
        else
        {
            sb++;
            oc = oc == 0 ? c.RC - 1 : oc - 1;
            down = 0;
            d = 0;
            orow = row; // (the cell in the same row always passes the loop condition: d = 0, row inside the image)
        }
    }
    __device__ __forceinline__ void next_direction(const ScanCfg& c) // a direction ended (break or loop condition false)
    {
        if (down == 0 && sb > 0)
        {
            down = 1;
            d = 1;
            orow = row + 1;
            if (!(orow < c.R && d <= c.max_col_steps))
                next_column(c);
        }
        else
            next_column(c);
    }
    // position on the first cell that passes the while-condition of cc.cpp:718-719, or finish
    __device__ __forceinline__ void start(const ScanCfg& c, const bool have, const int lc)
    {
        sb = 0, down = 0, d = 1, oc = lc, orow = row - 1;
        rooted = 0, parent = -1, nlinks = 0, overflow = 0, visits = 0, reach = 0;
        packed = 0;
        run = have;
        if (run && !(orow >= 0 && d <= c.max_col_steps))
            next_direction(c); // row 0 has nothing above it in its own column
    }
    template<bool MIRROR>
    __device__ __forceinline__ void step(const ScanCfg& c, const SP& p)
    {
        const float4 o = p.sc_rec[oc * c.R + orow];
        const unsigned char oign = p.ignored[oc * c.R + orow]; // (issued with the record: one round trip per visit)
        if (MIRROR)
        {
            visits++; // cc.cpp:725
            reach = sb;
        }
        if (ccm::absf(o.w - me.w) > mad) // cc.cpp:728: the inclination window is left
            next_direction(c);
        else
        {
            const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;
            if (!oign && dx * dx + dy * dy + dz * dz < c.maxd2) // (a cell without a return is ignored, and its x is NaN)
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
            if (rooted && c.stop_enabled && d >= c.stop_min) // cc.cpp:746-749
                next_direction(c);
            else
            {
                d++;
                orow = down ? orow + 1 : orow - 1;
                if (!(orow >= 0 && orow < c.R && d <= c.max_col_steps))
                    next_direction(c);
            }
        }
    }
};

// ---- long scans (round 6) -----------------------------------------------------------------------------------------------------------------------
// Nearly every scan is over after four visits, but a point that finds no neighbour (vegetation, spray, the edge of an object) visits every cell
// of its inclination window in up to max_steps_in_row + 1 columns: hundreds of visits (p99 331, max 840 on the bench's vegetation scenes; p99 5,
// max 200 on its street scene). In a wavefront whose lanes run one point each, 63 lanes then wait for that one: 5 % of the lanes did a visit per
// iteration on vegetation, 32 % on the street scene (tools/scan_model.py, from the oracle's visit counts). So a lane that is still scanning after
// SCAN_CAP visits hands its state to the stream's LONG-SCAN LIST (Planes::sl_rec) and k_scan2 goes on; k_scan2_long then runs those points with
// every lane busy — a lane whose point is done takes the next one from the list —, and k_scan2_epi runs the column epilogue of the columns that
// had such a point (k_scan2 does it at once for the others: 97 % of the columns on the street scene, 58 % on vegetation). Same visits in the same
// order per point, so the same results bit for bit. Only without Geometry::mirror_fields (the visit counts and the scan reach of the host mirror
// stay with the one-pass form).
constexpr int SL_CAP = 8192; // records of a stream's long-scan list per batch; a lane that finds it full finishes its scan where it is
struct ScanLongRec
{
    int ci;            // the point's cell (local column * rows + row)
    int oc;            // ring column of the next visit
    short orow, sb, d, needed;
    short parent;
    unsigned char down, rooted, nlinks, flags; // flags: 1 link overflow, 2 the scan must stop at the stream's first column (bound)
    float mad;
    unsigned long long packed;
};
static_assert(sizeof(ScanLongRec) == 40, "long-scan record");
// per stream: Planes::sl_ctl[4] = {records, next record to hand out, deferred columns, blocks of k_scan2_epi that are through}

#ifndef CC_SCAN_CAP
#define CC_SCAN_CAP 6
#endif
constexpr int SCAN_CAP = CC_SCAN_CAP; // visits a lane of k_scan2 spends on its point before it hands it to the long-scan list (SPLIT)

template<int RPL, bool MIRROR, bool SPLIT = false>
__global__ __launch_bounds__(64) void k_scan2(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot)
{
    static_assert(!(SPLIT && MIRROR), "the long-scan list does not carry the mirror's visit counts");
    const int s = first_stream + blockIdx.x;
    const int lane = lane_id();
    const StreamState* st = &states[s];
    if (st->error != 0 || st->batch[slot].seg_begin < 0 || st->batch[slot].mode != 0)
        return;
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols;
    ScanCfg sc;
    sc.R = R, sc.RC = RC;
    sc.maxd2 = g.max_distance_squared;
    sc.max_col_steps = cfg.max_steps_in_column;
    sc.stop_enabled = cfg.stop_after_association_enabled != 0;
    sc.stop_min = cfg.stop_after_association_min_steps;
    const int max_row_steps = cfg.max_steps_in_row;
    const int TC = SCAN_TILE_CELLS / (RPL * 64); // columns per tile: 4 at <= 64 rows, 2 at <= 128
    __shared__ unsigned short s_list[SCAN_TILE_CELLS];  // tile-local cell (column in tile * RPL * 64 + row) of every active point
    __shared__ short s_parent[SCAN_TILE_CELLS];         // (SPLIT: -3 = the point waits in the long-scan list)
    __shared__ unsigned char s_nlinks[SCAN_TILE_CELLS];
    __shared__ unsigned short s_visits[MIRROR ? SCAN_TILE_CELLS : 1];
    __shared__ unsigned char s_reach[MIRROR ? SCAN_TILE_CELLS : 1];
    __shared__ double s_fin[SCAN_TILE_CELLS];
    __shared__ unsigned long long s_links[SCAN_TILE_CELLS];
    int* const sl_ctl = P.sl_ctl + (size_t) s * 4;
    ScanLongRec* const sl_rec = (ScanLongRec*) P.sl_rec + (size_t) s * SL_CAP;
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
        unsigned deferred = 0; // SPLIT: bit tc = a point of the tile's column tc went to the long-scan list
        for (int base = 0; base < n_act; base += 64)
        {
            const bool have = base + lane < n_act;
            const int tl = have ? (int) s_list[base + lane] : 0;
            const int tc = tl / (RPL * 64), row = tl % (RPL * 64);
            const long long gc = gc0 + tc;
            int lc = lc0 + tc;
            lc = lc >= RC ? lc - RC : lc;
            const int ci = lc * R + row;
            ScanPoint q;
            q.me = make_float4(0.f, 0.f, 0.f, 0.f);
            q.mad = 0.f;
            q.row = row;
            double fin = 0.;
            q.needed = -1;
            if (have)
            {
                q.me = p.sc_rec[ci];
                q.mad = ccm::asinf_exact(cfg.max_distance / p.dist[ci]);
                fin = cell_caz(caz_base_of_rotation(rot0 + (cir0 + tc >= g.num_columns ? 1 : 0)), p.incaz[ci]) + (double) q.mad;
                q.needed = f2i_x86(__builtin_ceilf(q.mad / g.az_width));
                q.needed = q.needed < max_row_steps ? q.needed : max_row_steps;
            }
            // never look at columns older than the first column ever segmented (their planes are uninitialised)
            q.bound = (gc - first_column) <= (long long) max_row_steps + 1 ? first_lc : -1;
            q.start(sc, have, lc);
            bool pending = false;
            if (SPLIT)
            {
                for (int it = 0; it < g.scan_cap && __any(q.run); it++)
                    if (q.run)
                        q.template step<false>(sc, p);
                const unsigned long long m = __ballot(q.run);
                if (m)
                {
                    // the points that are still scanning go to the stream's long-scan list (one atomic per wavefront), as far as it has room
                    int lbase = 0;
                    if (lane == (int) __ffsll((long long) m) - 1)
                        lbase = atomicAdd(&sl_ctl[0], (int) __popcll(m));
                    lbase = __shfl(lbase, (int) __ffsll((long long) m) - 1);
                    const int idx = lbase + (int) __popcll(m & lanes_below());
                    if (q.run && idx < g.sl_cap)
                    {
                        ScanLongRec r;
                        r.ci = ci, r.oc = q.oc;
                        r.orow = (short) q.orow, r.sb = (short) q.sb, r.d = (short) q.d, r.needed = (short) q.needed;
                        r.parent = (short) q.parent;
                        r.down = (unsigned char) q.down, r.rooted = (unsigned char) q.rooted, r.nlinks = (unsigned char) q.nlinks;
                        r.flags = (unsigned char) ((q.overflow ? 1 : 0) | (q.bound >= 0 ? 2 : 0));
                        r.mad = q.mad;
                        r.packed = q.packed;
                        sl_rec[idx] = r;
                        pending = true;
                        q.run = false;
                    }
                    while (__any(q.run)) // (a full list: finish here)
                        if (q.run)
                            q.template step<false>(sc, p);
                }
                // which of the tile's columns have a waiting point
                for (int t2 = 0; t2 < TC; t2++)
                    if (__any(pending && tc == t2))
                        deferred |= 1u << t2;
            }
            else
            {
                while (__any(q.run))
                    if (q.run)
                        q.template step<MIRROR>(sc, p);
            }
            if (have)
            {
                s_parent[tl] = (short) (pending ? -3 : q.parent);
                s_nlinks[tl] = (unsigned char) (q.overflow ? 255 : q.nlinks);
                s_fin[tl] = fin;
                s_links[tl] = q.packed;
                if (MIRROR)
                {
                    s_visits[tl] = sat_u16(q.visits);
                    s_reach[tl] = (unsigned char) q.reach;
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
            const bool later = SPLIT && ((deferred >> tc) & 1u) != 0; // the column waits for k_scan2_long: k_scan2_epi finishes it
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
                    if (g.scan_stores_fin)
                        p.sc_fin[ci] = fin[k];
                    if (!(later && parent[k] == -3)) // (a waiting point's parent and links are written by k_scan2_long)
                    {
                        p.sc_parent[ci] = (int16_t) parent[k];
                        p.sc_nlinks[ci] = (uint8_t) nlinks[k];
                        if (nlinks[k] > 0)
                            p.sc_links[ci] = packed[k];
                    }
                    if (MIRROR)
                        p.sc_visits[ci] = s_visits[tl];
                }
            }
            if (later)
            {
                if (lane == 0)
                    p.sl_cols[atomicAdd(&sl_ctl[2], 1)] = lc;
            }
            else
                scan_column_epilogue<RPL, MIRROR>(p, R, lc, lane, parent, nlinks, fin, packed, reach);
        }
        wave_lds_fence(); // the tile's LDS arrays are rewritten by the next tile
    }
}

// k_scan2_long — the points k_scan2<.., SPLIT> handed over, one per lane, every lane busy: a lane whose point is done writes its results
// (Planes::sc_parent / sc_nlinks / sc_links of the point's cell) and takes the next record. grid = (streams, SCAN_LONG_BLOCKS), block = 64.
#ifndef CC_SCAN_LONG_BLOCKS
#define CC_SCAN_LONG_BLOCKS 16
#endif
constexpr int SCAN_LONG_BLOCKS = CC_SCAN_LONG_BLOCKS;
constexpr int SCAN_EPI_BLOCKS = 16;

template<int RPL>
__global__ __launch_bounds__(64) void k_scan2_long(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot, int* __restrict__ stat)
{
    const int s = first_stream + blockIdx.x;
    int* const sl_ctl = P.sl_ctl + (size_t) s * 4;
    int n = sl_ctl[0];
    if (n <= 0)
        return;
    n = n < g.sl_cap ? n : g.sl_cap;
    const StreamState* st = &states[s];
    const int lane = lane_id();
    const SP p = stream_ptrs(P, g, s);
    const ScanLongRec* const sl_rec = (ScanLongRec*) P.sl_rec + (size_t) s * SL_CAP;
    const int R = g.num_rows, RC = g.ring_cols;
    ScanCfg sc;
    sc.R = R, sc.RC = RC;
    sc.maxd2 = g.max_distance_squared;
    sc.max_col_steps = cfg.max_steps_in_column;
    sc.stop_enabled = cfg.stop_after_association_enabled != 0;
    sc.stop_min = cfg.stop_after_association_min_steps;
    const int first_lc = (int) (st->first_column % RC);
    ScanPoint q;
    q.run = false;
    q.me = make_float4(0.f, 0.f, 0.f, 0.f);
    q.mad = 0.f, q.row = 0, q.needed = 0, q.bound = -1, q.sb = 0, q.down = 0, q.d = 0, q.oc = 0, q.orow = 0;
    q.rooted = 0, q.parent = -1, q.nlinks = 0, q.overflow = 0, q.visits = 0, q.reach = 0, q.packed = 0ull;
    int ci = 0;
    int nvis = 0; // visits this lane made (the engine's automatic mode weighs the long scans by them)
    bool more = true; // (wave-uniform) the list may still have records nobody has taken
    for (;;)
    {
        const unsigned long long idle = __ballot(!q.run);
        if (more && idle)
        {
            int lbase = 0;
            if (lane == (int) __ffsll((long long) idle) - 1)
                lbase = atomicAdd(&sl_ctl[1], (int) __popcll(idle));
            lbase = __shfl(lbase, (int) __ffsll((long long) idle) - 1);
            const int idx = lbase + (int) __popcll(idle & lanes_below());
            if (!q.run && idx < n)
            {
                const ScanLongRec r = sl_rec[idx];
                ci = r.ci;
                q.me = p.sc_rec[ci];
                q.mad = r.mad;
                q.row = ci % R;
                q.needed = r.needed;
                q.bound = (r.flags & 2) ? first_lc : -1;
                q.sb = r.sb, q.down = r.down, q.d = r.d, q.oc = r.oc, q.orow = r.orow;
                q.rooted = r.rooted, q.parent = r.parent, q.nlinks = r.nlinks, q.overflow = r.flags & 1;
                q.packed = r.packed;
                q.run = true;
            }
            more = lbase + (int) __popcll(idle) < n;
        }
        if (!__any(q.run))
            break;
        for (int it = 0; it < 8; it++) // (a few visits between two looks at the list)
            if (q.run)
            {
                q.template step<false>(sc, p);
                nvis++;
                if (!q.run)
                {
                    p.sc_parent[ci] = (int16_t) q.parent;
                    p.sc_nlinks[ci] = (uint8_t) (q.overflow ? 255 : q.nlinks);
                    if (q.nlinks > 0)
                        p.sc_links[ci] = q.packed;
                }
            }
    }
    if (stat)
    {
        for (int off = 32; off > 0; off >>= 1)
            nvis += __shfl_down(nvis, off, 64);
        if (lane == 0 && nvis > 0)
            atomicAdd(&stat[1], nvis);
    }
}

// k_scan2_epi — the column epilogue of the columns that waited for k_scan2_long: their per-cell scan results are complete in the planes now.
// The last block through clears the stream's long-scan counters for the next batch. grid = (streams, SCAN_EPI_BLOCKS), block = 64.
template<int RPL>
__global__ __launch_bounds__(64) void k_scan2_epi(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot, int* __restrict__ stat)
{
    const int s = first_stream + blockIdx.x;
    int* const sl_ctl = P.sl_ctl + (size_t) s * 4;
    const int nd = sl_ctl[2];
    const int lane = lane_id();
    // what the engine's automatic mode looks at (cc_engine.hip: scan_split 2): long scans and columns of the batches scanned this way
    if (stat && blockIdx.y == 0 && lane == 0)
    {
        const StreamState* st = &states[s];
        if (st->error == 0 && st->batch[slot].seg_begin >= 0 && st->batch[slot].mode == 0)
        {
            atomicAdd(&stat[2], (int) (st->batch[slot].seg_end - st->batch[slot].acp_next));
        }
    }
    if (nd <= 0 && sl_ctl[0] <= 0)
        return;
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows;
    for (int i = blockIdx.y; i < nd; i += gridDim.y)
    {
        const int lc = p.sl_cols[i];
        const CazBase cb = caz_base_of_column(p.colg[lc], g.num_columns);
        int parent[RPL], nlinks[RPL];
        double fin[RPL];
        unsigned long long packed[RPL];
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            parent[k] = -2, nlinks[k] = 0, fin[k] = 0., packed[k] = 0ull;
            if (row < R)
            {
                const int ci = lc * R + row;
                parent[k] = p.sc_parent[ci];
                nlinks[k] = p.sc_nlinks[ci];
                fin[k] = cell_fin(cfg, p, ci, cb);
                if (nlinks[k] > 0)
                    packed[k] = p.sc_links[ci];
            }
        }
        scan_column_epilogue<RPL, false>(p, R, lc, lane, parent, nlinks, fin, packed, 0);
    }
    __threadfence();
    if (lane == 0 && atomicAdd(&sl_ctl[3], 1) == (int) gridDim.y - 1)
    {
        sl_ctl[0] = 0;
        sl_ctl[1] = 0;
        sl_ctl[2] = 0;
        sl_ctl[3] = 0;
    }
}
