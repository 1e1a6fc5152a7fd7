// Batch-parallel association + finished-cluster check (included by cc_kernels.h inside namespace cck, after cc_assoc3.h).
//
// k_assocb does what k_assoc3 / k_assoc_lds do (association bookkeeping cc.cpp:643-696 and 773-835, finished-cluster check :837-974, publish
// bookkeeping :1035-1092), but not as a walk over the columns: a block of AB_WAVES wavefronts takes a GROUP of up to 64 columns of one stream
// at a time and nothing in it is serial per column.
//
//   1  tree of every point   k_scan left, per point, where its chain of same-column parents ends (sc_term): a new root of the column or a cell
//                            of an earlier column. Inside the group that is a forest of pointers, resolved by pointer jumping in an LDS ring
//                            of per-cell tree slots (<= 7 rounds for 64 columns, instead of one dependent look-up per column).
//   2  records               per column and tree that receives points: count and largest finished_at contribution (cc.cpp:666-670). What the
//                            finished-cluster check needs of them is ONE 64-bit word per tree: bit j = "this tree alone keeps its cluster
//                            unfinished at column j of the group" (its largest finished_at so far > the column's smallest azimuth, :884-885).
//                            A record (column c, value f) contributes ballot(lane >= c && f > min_az[lane]) — one compare over the lanes, one
//                            LDS atomic OR.
//   3  timeline (one wave)   lanes = trees. A cluster is unfinished at column j iff the OR of its trees' words has bit j set; it is finished at
//                            the first eligible column where it has not. Tree links (cc.cpp:675-696) are the only thing that changes clusters:
//                            the few links of a group that join two different trees are applied in column order, the columns between two such
//                            events are one "epoch" evaluated with a single segmented OR. Ids by rank of (finish column, oldest tree),
//                            first-unpublished column per column as the oldest tree still listed (cc.cpp:944-959), events by prefix sums.
//   4  commit                tree roots of the group's cells, finished trees, compaction of the tree table, remap of the slot ring.
//
// Exactness. The group is evaluated as if every first accepted candidate rooted its point (k_scan's assumption) and nothing in it touched a
// finished tree. Every way the reference's sequential semantics can differ is DETECTED before anything is committed — a point whose chain ends in
// a finished tree or in no tree (attach refused, cc.cpp:658), a tree that receives a point after its cluster finished inside the group, a tree or
// cluster that could span a rotation (:657, :913-924), a candidate taken from a column older than the first unpublished one (:762-763), link
// list overflow, more trees / links than the group's lanes — and then the kernel stops in front of that group: the exact serial kernel (k_assoc3,
// launched behind it) continues from there to the end of the batch. A link into a cluster that finished earlier in the group is refused exactly
// like the reference refuses it (no rescan needed). Rounds whose smallest azimuth equals the previous round's finish nothing,
// handled inline. Same results as the serial kernels, bit for bit (every parity test runs with this kernel in front; option "assoc_batch" = 0
// takes it out).
#pragma once

#ifndef CC_AB_WAVES
#define CC_AB_WAVES 15
#endif
#ifndef CC_AB_CPW
#define CC_AB_CPW 4
#endif
constexpr int AB_WAVES = CC_AB_WAVES;      // worker wavefronts (columns); one more wavefront runs the timeline
constexpr int AB_THREADS = 64 * (AB_WAVES + 1);
constexpr int AB_CPW = CC_AB_CPW;          // columns per worker wavefront and group
constexpr int AB_G = AB_WAVES * AB_CPW;    // columns per group (<= 64 = lanes of the timeline wave)
constexpr int AB_RING = 128;               // columns of the slot ring (power of two >= AB_G + WIN_COLS)
constexpr int AB_TREES = 64;               // trees a group can see (unfinished at its start + born inside) = lanes of the timeline wave
constexpr int AB_EVENTS = 64;              // links between different trees per group
constexpr int AB_NONE = -100, AB_DEAD = -101; // ring entries: cell without a tree / tree finished
// why a launch handed the rest of its batch to the serial kernel (StreamState::batch_bail_reason)
enum
{
    AB_BAIL_TREES = 1,    // more unfinished trees than lanes
    AB_BAIL_LINKS = 2,    // link list overflow in k_scan, or more tree links in a group than lanes
    AB_BAIL_ROTATION = 3, // a tree / cluster could reach the one-rotation limits
    AB_BAIL_DEAD = 4,     // a chain of parents ends in a finished tree or in a cell without a tree
    AB_BAIL_LATE = 5,     // a tree receives a point after its cluster finished inside the group
    AB_BAIL_REACH = 6,    // a candidate from a column older than the first unpublished one
};
static_assert(AB_G <= 64 && AB_THREADS <= 1024 && AB_RING >= AB_G + WIN_COLS, "group geometry");

struct AbTrees
{
    // persistent over the groups: the unfinished trees in creation order (the reference's sc_unfinished_point_trees_)
    int cell[AB_TREES];
    long long gcol[AB_TREES];
    unsigned long long fin[AB_TREES]; // bits of finished_at_continuous_azimuth_angle (non-negative double)
    long long last[AB_TREES];         // last global column that attached a point
    unsigned pts[AB_TREES];
    int comp[AB_TREES];               // cluster = smallest list position of its trees
    // per group
    unsigned long long g_alive[AB_TREES]; // bit j: the tree alone keeps its cluster unfinished at column j of the group
    unsigned long long g_fin[AB_TREES];
    unsigned g_pts[AB_TREES];
    int g_last[AB_TREES];             // last column of the group (relative) that attached a point, -1 none
    int birth[AB_TREES];              // column of the group (relative) the tree starts in, -1: older
    int g_cell[AB_TREES];             // root cell of the trees born in the group
    unsigned long long k_or[AB_TREES];
    unsigned k_pts[AB_TREES];
    long long k_max[AB_TREES];
    unsigned k_cid[AB_TREES];
    long long t_gcol[AB_TREES];
    int remap[AB_TREES];
    int cell_old[AB_TREES];
    unsigned ev[AB_EVENTS];           // column << 16 | tree a << 8 | tree b
    int col_base[64];               // new roots of the group's earlier columns
    int col_info[64];
    double min_az[64];
    int col_ncl[64];
    int col_ebase[64];
    int col_fix[64];                // >= 0: ring column of the first unpublished column while the column was associated (visit counts are re-taken)
    int n_ev;
    int ncols;
    int nborn;
    int bail;      // the group in work cannot be taken (nothing of it is committed)
    int next_bail; // the kernel stops in front of the next group
    int any_finished;
    int jflag[8];  // pointer jumping: round r left pointers unresolved
    int next_info[64]; // column summaries of the next group (link flags for the input prefetch)
    // links already reported for the column a worker wavefront is at: bit b of pair[wave][a] = (column, tree a, tree b) is in the event list. Where two
    // trees of one object meet, every point along the seam reports the same pair: without this filter such a group had more events than the
    // timeline has lanes (AB_EVENTS) and the kernel stopped — the only stop ordinary streams still produced
    unsigned long long pair[AB_WAVES][64];
};

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the global loads requested a group
// ahead and for the root-plane stores — none of which another wavefront of the block ever reads.
__device__ __forceinline__ void ab_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// inclusive prefix sum over the 64 lanes by DPP (row_shr 1/2/4/8, row_bcast 15/31): no LDS round trips
__device__ __forceinline__ int wave_incl_add_i32(int v)
{
    v += dpp_mov_i32<0x111, 0xf>(0, v);
    v += dpp_mov_i32<0x112, 0xf>(0, v);
    v += dpp_mov_i32<0x114, 0xf>(0, v);
    v += dpp_mov_i32<0x118, 0xf>(0, v);
    v += dpp_mov_i32<0x142, 0xa>(0, v);
    v += dpp_mov_i32<0x143, 0xc>(0, v);
    return v;
}
// the value of the lane below (wave_shr:1); lane 0 keeps its own
__device__ __forceinline__ long long wave_shr1_i64(long long v)
{
    return dpp_mov_i64<0x138, 0xf>(v, v);
}
__device__ __forceinline__ double wave_shr1_f64(double v)
{
    return __longlong_as_double(wave_shr1_i64(__double_as_longlong(v)));
}

template<int RPL>
__global__ __launch_bounds__(AB_THREADS) void k_assocb(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot,
                                                       int* __restrict__ bail_count)
{
    const int s = first_stream + blockIdx.x;
    const int lane = lane_id();
    const int wave = uniform_i32((int) (threadIdx.x >> 6));
    StreamState* st = &states[s];
    if (st->error != 0 || st->batch[slot].seg_begin < 0 || st->assoc_mode != 0 || st->batch[slot].mode != 0 ||
        st->batch[slot].acp_next >= st->batch[slot].seg_end)
        return;
    AssocCtx c;
    c.p = stream_ptrs(P, g, s);
    const SP& p = c.p;
    const int R = c.R = g.num_rows, NC = c.NC = g.num_columns, RC = c.RC = g.ring_cols;
    c.az_width = g.az_width;
    c.maxd2 = g.max_distance_squared;
    c.max_steps_in_row = cfg.max_steps_in_row;
    c.max_steps_in_column = cfg.max_steps_in_column;
    c.stop_enabled = cfg.stop_after_association_enabled;
    c.stop_min_steps = cfg.stop_after_association_min_steps;
    const int tree_limit = g.lds_tree_limit < AB_TREES ? g.lds_tree_limit : AB_TREES;
    int n_unf = st->n_unfinished;
    if (n_unf > tree_limit || cfg.max_steps_in_row > WIN_COLS - 2)
    {
        // the serial kernels decide (LDS pool / global memory). Counted like a stop in front of the first group: the host keeps ONE serial block
        // per stream while this counter moves (cc_engine.hip: bail_seen / bail_cooldown) — with the two sweeping blocks it launches behind an idle
        // batch-parallel kernel, a dense scene (65 .. 256 unfinished trees on many streams) would be associated by two blocks, stream after stream
        if (threadIdx.x == 0)
        {
            st->batch_bails += 1ull;
            st->batch_bail_reason[AB_BAIL_TREES] += 1ull;
            if (bail_count)
                atomicAdd(bail_count, 1);
        }
        return;
    }

#ifndef CC_AB_PRIO
#define CC_AB_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(CC_AB_PRIO); // latency-bound (barriers, LDS round trips): win issue arbitration against co-resident throughput kernels
    __shared__ AbTrees T;
    __shared__ short ring[AB_RING * WAVE * RPL];
    __shared__ int s_nunf;

    const long long col_begin = st->batch[slot].acp_next, col_end = st->batch[slot].seg_end, first_column = st->first_column;
    long long first_unpub = st->first_unpublished, ring_start = st->ring_start;
    unsigned long long cluster_counter = st->cluster_counter;
    double last_min_az = st->last_round_min_az;
    unsigned long long cells_published = st->cells_published, clusters_finished = st->clusters_finished;
    unsigned long long alias_rounds = st->stamp_alias_rounds;
    int n_events = st->n_events;

    // ---- the persistent tree state (global planes indexed by root cell) -> LDS, list position = slot ---------------------------------------
    if ((int) threadIdx.x < n_unf)
    {
        const int i = threadIdx.x;
        const int cell = p.ulist[i];
        const long long tg = p.colg[cell / R];
        T.cell[i] = cell;
        T.gcol[i] = tg;
        T.fin[i] = (unsigned long long) __double_as_longlong(p.t_fin[cell]);
        T.last[i] = tg + (long long) p.t_width[cell] - 1;
        T.pts[i] = p.t_pts[cell];
        T.comp[i] = p.t_pos[p.t_uf[cell]];
    }
    {
        // slot ring of the WIN_COLS columns before col_begin: two dependent gathers per cell (root plane, then the tree planes at the root)
        for (int i = threadIdx.x; i < WIN_COLS * R; i += AB_THREADS)
        {
            const int back = i / R + 1, row = i - (back - 1) * R;
            const long long gcx = col_begin - back;
            int v = AB_NONE;
            if (gcx >= first_column && gcx >= 0 && first_column >= 0)
            {
                const int r = p.root[(int) (gcx % RC) * R + row];
                if (r >= 0)
                    v = p.t_finished[r] ? AB_DEAD : -1 - p.t_pos[r];
            }
            ring[(int) (gcx & (AB_RING - 1)) * R + row] = (short) v;
        }
    }
    __syncthreads();
    if (wave == 0)
    {
        // union-find parents -> cluster representative = the smallest list position of the set (the serial kernels that may have left this
        // state number their trees by ids from a free ring: the root of a set is its smallest ID, which need not be its oldest tree)
        int rep = lane < n_unf ? T.comp[lane] : 0;
        for (int it = 0; it < 6; it++)
        {
            const int r2 = __shfl(rep, rep);
            rep = r2;
        }
        T.remap[lane] = AB_TREES;
        wave_lds_fence();
        if (lane < n_unf)
            atomicMin(&T.remap[rep], lane);
        wave_lds_fence();
        if (lane < n_unf)
            T.comp[lane] = lds_ld(&T.remap[rep]);
        if (lane == 0 && st->batch[slot].pub_begin < 0)
            st->batch[slot].pub_begin = first_unpub; // first association kernel of this pass
    }
    __syncthreads();

#ifdef CC_AB_STATS
    unsigned long long ab_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ab_mark = __builtin_amdgcn_s_memtime();
    const unsigned long long ab_t0 = ab_mark;
#define AB_PH(i)                                                     \
    {                                                                \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
        ab_t[i] += n_ - ab_mark;                                     \
        ab_mark = n_;                                                \
    }
#else
#define AB_PH(i)
#endif
    // CC_AB_STATS_W (with CC_AB_STATS): the counters are taken by worker wavefront 1 instead of the timeline wavefront
#if defined(CC_AB_STATS) && defined(CC_AB_STATS_W)
#undef AB_PH
#define AB_PH(i)
#define AB_PHW(i)                                                    \
    if (wave == 1)                                                   \
    {                                                                \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
        ab_t[i] += n_ - ab_mark;                                     \
        ab_mark = n_;                                                \
    }
#else
#define AB_PHW(i)
#endif
    long long gc0 = col_begin;
    int lc0 = (int) (col_begin % RC);
    unsigned long long batch_cols = 0;
    bool bailed = false;
    const double mz_inf = 1.7976931348623157e308;
    if (threadIdx.x == 0)
        T.bail = 0;

    // Two roles, one barrier schedule per group: B1 (pointers written) - one barrier per pointer-jumping round - B2 (records, links) - B3
    // (timeline done, group committed, next header in place). The timeline wavefront and the workers run separate loops (separate register
    // allocations: the workers hold a group of prefetched inputs, the timeline wave ~60 per-tree / per-column values) that meet at these barriers.
    if (wave == 0)
    {
        // =============================================================================================== timeline wavefront
        // column summaries of the NEXT group in registers (lanes = columns), requested a whole group ahead
        int nx_info = 0;
        double nx_maz = mz_inf;
        auto load_next_info = [&](const long long g0, const int l0)
        {
            nx_info = 0;
            nx_maz = mz_inf;
            if (lane < AB_G && g0 + lane < col_end)
            {
                int lcj = l0 + lane;
                lcj = lcj >= RC ? lcj - RC : lcj;
                nx_info = p.col_info[lcj];
                nx_maz = p.colminaz[lcj];
            }
        };
        // group header: how many columns the group takes (the trees it sees must fit the lanes), what stops the kernel in front of it;
        // then the summaries of the group after it are requested
        auto group_header = [&](const long long g0, const int l0, const int n_before)
        {
            const int info = nx_info;
            const double maz = nx_maz;
            const bool valid = lane < AB_G && g0 + lane < col_end;
            const int cnt = info & 0xff;
            const int incl = wave_incl_add_i32(cnt);
            const unsigned long long okm = __ballot(valid && n_before + incl <= tree_limit);
            const int ncols = ~okm ? __builtin_ctzll(~okm) : 64;
            const unsigned long long cm = ncols >= 64 ? ~0ull : ((1ull << ncols) - 1ull);
            int bail = (ncols == 0 && g0 < col_end) ? AB_BAIL_TREES : 0;
            if (__ballot(((info >> 8) & 1) != 0) & cm)
                bail = AB_BAIL_LINKS; // a point with more link candidates than k_scan records (cc.cpp:693-694 would see them all)
            // no tree or cluster of this group can reach the one-rotation limits (cc.cpp:657, 913-924) while the oldest unfinished tree is
            // less than a rotation behind the group's last column
            if (n_before > 0 && ncols > 0 && (g0 + ncols - lds_ld(&T.gcol[0])) >= NC)
                bail = AB_BAIL_ROTATION;
            T.col_base[lane] = incl - cnt;
            T.col_info[lane] = info;
            T.min_az[lane] = maz;
            T.col_ncl[lane] = 0;
            T.g_alive[lane] = 0ull;
            T.g_fin[lane] = 0ull;
            T.g_pts[lane] = 0u;
            T.g_last[lane] = -1;
            T.birth[lane] = -1;
            const int nb = __builtin_amdgcn_readlane(incl, ncols > 0 ? ncols - 1 : 0);
            if (lane < 8)
                T.jflag[lane] = 0;
            if (lane == 0)
            {
                T.n_ev = 0;
                T.ncols = ncols;
                T.nborn = ncols > 0 ? nb : 0;
                T.next_bail = bail;
            }
            int l1 = l0 + ncols;
            l1 = l1 >= RC ? l1 - RC : l1;
            load_next_info(g0 + ncols, l1);
        };
        load_next_info(gc0, lc0);
        T.next_info[lane] = nx_info;
        ab_barrier(); // P0a: link flags of the first group for the workers' prefetch
        group_header(gc0, lc0, n_unf);
        ab_barrier(); // P0b
        if (T.next_bail)
        {
            bailed = true;
            if (lane == 0)
                T.bail = T.next_bail;
        }
        while (gc0 < col_end && !bailed)
        {
            AB_PH(0)
            const int ncols = T.ncols;
            const int nborn = T.nborn;
            const double mz = T.min_az[lane]; // lanes = columns of the group
            // what the trees that are older than the group contribute to the finished-cluster check of its columns
            for (int t = 0; t < n_unf; t++)
            {
                const double f = __longlong_as_double((long long) lds_ld(&T.fin[t]));
                const unsigned long long m = __ballot(f > mz);
                if (lane == 0)
                    T.g_alive[t] = m;
            }
            ab_barrier(); // B1
            AB_PH(1)
            for (int r = 0; r < 8; r++)
            {
                ab_barrier();
                if (!T.jflag[r])
                    break;
            }
            AB_PH(2)
            T.next_info[lane] = nx_info; // (requested a group ago: the workers' prefetch reads the link flags of the next group's columns)
            ab_barrier(); // B2
            AB_PH(4)
            if (T.bail || T.n_ev > AB_EVENTS)
            {
                if (lane == 0 && !T.bail)
                    T.bail = AB_BAIL_LINKS;
                bailed = true;
                break;
            }
            // ================================================================================== 3: timeline (lanes = trees / columns)
            {
            const int n = n_unf + nborn;
            const bool is_t = lane < n;
            const bool old = lane < n_unf;
            const int birth = is_t ? T.birth[lane] : -1;
            const int cell = old ? T.cell[lane] : (is_t ? T.g_cell[lane] : 0);
            const long long tg = old ? T.gcol[lane] : gc0 + birth;
            const unsigned long long alive_w = is_t ? T.g_alive[lane] : ~0ull;
            const int g_last = is_t ? T.g_last[lane] : -1;
            int comp = old ? T.comp[lane] : lane;
            int fc = 64; // column of the group (relative) whose check finishes the tree's cluster; 64 = not in this group
            T.t_gcol[lane] = tg;
            // rounds that can finish something: not the ones whose smallest azimuth equals the previous round's (the BFS of cc.cpp:854 then
            // meets its own visited stamp everywhere)
            const double prev_az = wave_shr1_f64(mz);
            const bool alias = lane < ncols && mz == (lane == 0 ? last_min_az : prev_az);
            const unsigned long long colmask = ncols >= 64 ? ~0ull : ((1ull << ncols) - 1ull);
            const unsigned long long alias_m = __ballot(alias);
            const unsigned long long elig = ~alias_m & colmask;
            const int nev = T.n_ev;
            const unsigned evw = lane < nev ? T.ev[lane] : 0u;
            const int evcol = (int) (evw >> 16), eva = (int) ((evw >> 8) & 0xff), evb = (int) (evw & 0xff);
            bool ev_done = lane >= nev;
            bool ev_made = false;
            int epoch = 0;
            while (true)
            {
                int m = uniform_i32(wave_min_i32(ev_done ? 64 : evcol));
                m = m < ncols ? m : ncols;
                if (m > epoch)
                {
                    const unsigned long long rm = (m >= 64 ? ~0ull : ((1ull << m) - 1ull)) & ~((1ull << epoch) - 1ull);
                    const bool live = is_t && fc == 64;
                    T.k_or[lane] = 0ull;
                    wave_lds_fence();
                    if (live)
                        atomicOr(&T.k_or[comp], alive_w);
                    wave_lds_fence();
                    const unsigned long long o = live ? lds_ld(&T.k_or[comp]) : ~0ull;
                    const unsigned long long cand = ~o & elig & rm;
                    if (live && cand)
                        fc = __builtin_ctzll(cand);
                    wave_lds_fence();
                }
                if (m >= ncols)
                    break;
                // the links made in column m, in any order (a union is a union); a link to or from a finished tree is refused (cc.cpp:688-690)
                unsigned long long em = __ballot(!ev_done && evcol == m);
                while (em)
                {
                    const int k = __builtin_ctzll(em);
                    em &= em - 1ull;
                    const int ea = __builtin_amdgcn_readlane(eva, k), eb = __builtin_amdgcn_readlane(evb, k);
                    const int fa = __builtin_amdgcn_readlane(fc, ea), fb = __builtin_amdgcn_readlane(fc, eb);
                    if (fa == 64 && fb == 64)
                    {
                        const int ca = __builtin_amdgcn_readlane(comp, ea), cb = __builtin_amdgcn_readlane(comp, eb);
                        const int lo = ca < cb ? ca : cb, hi = ca < cb ? cb : ca;
                        if (comp == hi)
                            comp = lo;
                        if (lane == k)
                            ev_made = true;
                    }
                }
                ev_done = ev_done || evcol == m;
                epoch = m;
            }
            // a tree that receives a point after its cluster finished: the reference refuses the attach and scans on (cc.cpp:658)
            int tbad = (is_t && fc < 64 && g_last > fc) ? AB_BAIL_LATE : 0;

            // ---- finished clusters: points, extent, ids in the order the reference's BFS meets them (column, then oldest tree)
            const unsigned pts_t = (old ? T.pts[lane] : 0u) + (is_t ? T.g_pts[lane] : 0u);
            const long long last_old = old ? T.last[lane] : -1;
            const long long last_t = (g_last >= 0 && gc0 + g_last > last_old) ? gc0 + g_last : last_old;
            const unsigned long long fin_old = old ? T.fin[lane] : 0ull;
            const unsigned long long fin_g = is_t ? T.g_fin[lane] : 0ull;
            const unsigned long long fin_t = fin_g > fin_old ? fin_g : fin_old;
            T.k_pts[lane] = 0u;
            T.k_max[lane] = -1;
            T.k_cid[lane] = 0u;
            wave_lds_fence();
            const bool fin_here = is_t && fc < 64;
            if (fin_here)
            {
                atomicAdd(&T.k_pts[comp], pts_t);
                atomicMax(&T.k_max[comp], last_t);
            }
            wave_lds_fence();
            const bool is_rep = fin_here && comp == lane;
            const unsigned cpts = lds_ld(&T.k_pts[lane]);
            const long long cmax = lds_ld(&T.k_max[lane]);
            const bool has_id = is_rep && cpts > 5u; // cc.cpp:936
            const int key = has_id ? fc * 64 + lane : 0x7fffffff;
            int rank = 0, rank_col = 0;
            for (int i = 0; i < n; i++)
            {
                const int ki = __builtin_amdgcn_readlane(key, i);
                rank += ki < key ? 1 : 0;
                rank_col += (ki < key && (ki >> 6) == fc) ? 1 : 0;
            }
            const unsigned cid = (unsigned) (cluster_counter + (unsigned long long) rank);
            const int n_ids = __popcll(__ballot(has_id));
            if (has_id)
            {
                T.k_cid[lane] = cid;
                atomicAdd(&T.col_ncl[fc], 1);
            }
            wave_lds_fence();
            const unsigned cid_t = fin_here ? lds_ld(&T.k_cid[comp]) : 0u;

            // ---- per column (lanes = columns): the oldest tree still listed when the column's check ends (cc.cpp:944-959)
            int idx = -1;
            for (int i = n - 1; i >= 0; i--)
            {
                const int fi = __builtin_amdgcn_readlane(fc, i);
                if (fi >= lane)
                    idx = i;
            }
            const long long gcj = gc0 + lane;
            long long mc = gcj + 1;
            bool listed = false;
            if (idx >= 0)
            {
                const int bi = T.birth[idx];
                if (bi <= lane)
                {
                    mc = T.t_gcol[idx];
                    listed = true;
                }
            }
            const long long mc_prev = wave_shr1_i64(mc);
            const long long fu = lane == 0 ? first_unpub : mc_prev; // first unpublished column while column j is associated
            const int info = T.col_info[lane];
            const int reach = (info >> 16) & 0xff;
            if (lane < ncols && (mc < fu || gcj - reach < fu))
                tbad = AB_BAIL_REACH;
            // (mirror) k_scan's visit counts (Point::number_of_visited_neighbors, cc.cpp:725) are only right if no scan LOOKED past the first
            // unpublished column (cc.cpp:762-763): where one did — without accepting anything there — the counts are taken again in 4b
            int fix = -1;
            if (g.mirror_fields && lane < ncols && gcj - ((info >> 24) & 0x7f) < fu)
            {
                int lcf = lc0 + lane - (int) (gcj - fu);
                lcf = lcf >= RC ? lcf - RC : lcf;
                fix = lcf < 0 ? lcf + RC : lcf;
            }
            T.col_fix[lane] = fix; // cc.cpp:762-763: the live scan would have stopped earlier (or the bookkeeping error of :1072-1075: the serial kernel reports it)
            const int ncl = T.col_ncl[lane];
            const int per_col = lane < ncols ? 2 + ncl : 0;
            const int eincl = wave_incl_add_i32(per_col);
            T.col_ebase[lane] = n_events + eincl - per_col;
            wave_lds_fence();

            if (__any(tbad))
            {
                if (lane == 0)
                    T.bail = __any(tbad == AB_BAIL_LATE) ? AB_BAIL_LATE : AB_BAIL_REACH;
            }
            else
            {
                // ============================================================================== 4a: commit (still wave 0)
                if (g.record_events)
                {
                    if (lane < ncols)
                    {
                        const int e0 = T.col_ebase[lane];
                        if (e0 < g.event_capacity)
                        {
                            cc_event e;
                            e.type = CC_EV_GROUND_COLUMN;
                            e.stream = s;
                            e.a = gcj;
                            e.b = gcj;
                            e.c = 0;
                            e.d = 0;
                            e.column = gcj;
                            p.events[e0] = e;
                        }
                        const int e1 = e0 + 1 + ncl;
                        if (e1 < g.event_capacity)
                        {
                            cc_event e;
                            e.type = CC_EV_PUBLISH_COLUMNS;
                            e.stream = s;
                            e.a = fu;
                            e.b = mc - 1;
                            e.c = 0;
                            e.d = 0;
                            e.column = gcj;
                            p.events[e1] = e;
                        }
                    }
                    if (has_id)
                    {
                        const int e2 = T.col_ebase[fc] + 1 + rank_col;
                        if (e2 < g.event_capacity)
                        {
                            cc_event e;
                            e.type = CC_EV_CLUSTER;
                            e.stream = s;
                            e.a = tg;
                            e.b = cmax;
                            e.c = cid;
                            e.d = cpts;
                            e.column = gc0 + fc;
                            p.events[e2] = e;
                        }
                    }
                    n_events += __builtin_amdgcn_readlane(eincl, 63);
                }
                const long long new_unpub = lane_i64(mc, ncols - 1);
                cells_published += (unsigned long long) (new_unpub - first_unpub) * (unsigned long long) R;
                first_unpub = new_unpub;
                ring_start = first_unpub - NC > 0 ? first_unpub - NC : 0;
                cluster_counter += (unsigned long long) n_ids;
                clusters_finished += (unsigned long long) n_ids;
                alias_rounds += (unsigned long long) __popcll(alias_m & __ballot(listed)); // (counted like the serial kernels: only while trees are listed)
                last_min_az = lane_f64(mz, ncols - 1);
                // finished trees leave the list: what k_publish and the host mirror read of them
                if (fin_here)
                {
                    p.t_finished[cell] = 1;
                    p.t_cid[cell] = cid_t;
                    if (g.mirror_fields)
                    {
                        p.t_fin[cell] = __longlong_as_double((long long) fin_t);
                        p.t_pts[cell] = pts_t;
                        p.t_width[cell] = (unsigned) (last_t - tg) + 1u;
                    }
                }
                if (g.mirror_fields) // Point::associated_trees of both roots (cc.cpp:693-694)
                {
                    const int cell_a = __shfl(cell, eva), cell_b = __shfl(cell, evb);
                    if (ev_made)
                        log_link(g, st, p.link_log, cell_a, cell_b);
                }
                // stable compaction of the list
                const bool surv = is_t && fc == 64;
                const unsigned long long sm = __ballot(surv);
                const int np = __popcll(sm & lanes_below());
                T.remap[lane] = surv ? np : -1;
                T.cell_old[lane] = cell;
                wave_lds_fence();
                if (surv)
                {
                    T.cell[np] = cell;
                    T.gcol[np] = tg;
                    T.fin[np] = fin_t;
                    T.last[np] = last_t;
                    T.pts[np] = pts_t;
                    T.comp[np] = lds_ld(&T.remap[comp]);
                }
                n_unf = __popcll(sm);
                if (lane == 0)
                {
                    T.any_finished = n_unf != n ? 1 : 0;
                    s_nunf = n_unf;
                }
                wave_lds_fence();
                // the next group's header: by the time the other wavefronts have written this group's roots it is in place
                int l1 = lc0 + ncols;
                l1 = l1 >= RC ? l1 - RC : l1;
                group_header(gc0 + ncols, l1, n_unf);
            }
            }
            wave_lds_fence();
            ab_barrier(); // B3
            AB_PH(5)
            if (T.bail)
            {
                bailed = true;
                break;
            }
#ifdef CC_AB_STATS
            ab_t[7]++;
#endif
            batch_cols += (unsigned long long) ncols;
            gc0 += ncols;
            lc0 += ncols;
            lc0 = lc0 >= RC ? lc0 - RC : lc0;
            if (T.next_bail)
            {
                bailed = true;
                if (lane == 0)
                    T.bail = T.next_bail;
            }
        }
        if (lane == 0)
            s_nunf = n_unf;
    }
    else
    {
        // =============================================================================================== worker wavefronts
        const int w = wave - 1;
        // inputs of a group, requested a group ahead (the link words only where the column has links: T.next_info)
        int pf_par[AB_CPW][RPL], pf_term[AB_CPW][RPL], pf_nl[AB_CPW][RPL];
        double pf_fin[AB_CPW][RPL];
        unsigned long long pf_lk[AB_CPW][RPL];
        auto prefetch_inputs = [&](const long long g0, const int l0)
        {
#pragma unroll
            for (int q = 0; q < AB_CPW; q++)
            {
                const int cidx = w + q * AB_WAVES;
                const bool col_links = ((T.next_info[cidx] >> 8) & 3) != 0; // (bit 1: points with links, bit 0: points whose link list overflowed)
                int lc = l0 + cidx;
                lc = lc >= RC ? lc - RC : lc;
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    pf_par[q][k] = -2;
                    pf_term[q][k] = -1;
                    pf_fin[q][k] = 0.;
                    pf_nl[q][k] = 0;
                    pf_lk[q][k] = 0ull;
                    if (row < R && g0 + cidx < col_end)
                    {
                        const int ci = lc * R + row;
                        pf_par[q][k] = p.sc_parent[ci];
                        pf_term[q][k] = p.sc_term[ci];
                        pf_fin[q][k] = p.sc_fin[ci];
                        if (col_links)
                        {
                            pf_nl[q][k] = p.sc_nlinks[ci];
                            pf_lk[q][k] = p.sc_links[ci]; // (stale where the point has no links: never looked at)
                        }
                    }
                }
            }
        };
        ab_barrier(); // P0a
        prefetch_inputs(gc0, lc0);
        ab_barrier(); // P0b
        if (T.next_bail)
            bailed = true;
        while (gc0 < col_end && !bailed)
        {
            const int ncols = T.ncols;
            const double mz = T.min_az[lane]; // lanes = columns of the group
            // ================================================================================== 1: initial pointers
            int a[AB_CPW][RPL]; // ring value of the cell: >= 0 pointer (ring index), < 0 resolved (-1 - slot, AB_NONE, AB_DEAD)
#pragma unroll
            for (int q = 0; q < AB_CPW; q++)
            {
                const int cidx = w + q * AB_WAVES;
#pragma unroll
                for (int k = 0; k < RPL; k++)
                    a[q][k] = AB_NONE;
                if (cidx < ncols)
                {
                    int lc = lc0 + cidx;
                    lc = lc >= RC ? lc - RC : lc;
                    const long long gc = gc0 + cidx;
                    const int base = n_unf + T.col_base[cidx];
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                    {
                        const int row = k * 64 + lane;
                        if (row < R)
                        {
                            const int ci = lc * R + row;
                            const int par = pf_par[q][k];
                            const int term = pf_term[q][k];
                            int v = AB_NONE;
                            if (par >= -1)
                            {
                                if (term >= 256)
                                    v = (int) ((gc - (term >> 8)) & (AB_RING - 1)) * R + (term & 0xff);
                                else if (term >= 0)
                                    v = -1 - (base + term);
                                if (par == -1)
                                {
                                    const int sl = base + term;
                                    T.g_cell[sl] = ci;
                                    T.birth[sl] = cidx;
                                }
                            }
                            a[q][k] = v;
                            ring[(int) (gc & (AB_RING - 1)) * R + row] = (short) v;
                        }
                    }
                }
            }
            AB_PHW(0)
            ab_barrier(); // B1
            AB_PHW(1)
            // ---------------------------------------------------------------------------------- pointer jumping (one barrier per round)
            for (int r = 0; r < 8; r++)
            {
                int pending = 0;
#pragma unroll
                for (int q = 0; q < AB_CPW; q++)
                {
                    const int cidx = w + q * AB_WAVES;
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                        if (a[q][k] >= 0)
                        {
                            int b = ring[a[q][k]];
                            if (b >= 0)
                                b = ring[b]; // (two hops per round: a barrier costs more than an LDS round trip)
                            a[q][k] = b;
                            ring[(int) ((gc0 + cidx) & (AB_RING - 1)) * R + k * 64 + lane] = (short) b;
                            pending |= b >= 0 ? 1 : 0;
                        }
                }
                if (__any(pending) && lane == 0)
                    T.jflag[r] = 1;
                ab_barrier();
                if (!T.jflag[r])
                    break;
            }
            AB_PHW(2)
            // ================================================================================== 2: records, alive words, links
        int bad = 0;
#pragma unroll
        for (int q = 0; q < AB_CPW; q++)
        {
            const int cidx = w + q * AB_WAVES;
            if (cidx >= ncols)
                continue;
            int sl[RPL];
            unsigned long long am = 0ull;
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                sl[k] = -1;
                if (pf_par[q][k] >= -1) // an active point
                {
                    if (a[q][k] <= AB_NONE)
                        bad = 1; // its chain ends in a finished tree (attach refused, cc.cpp:658) or in a cell without a tree
                    else
                        sl[k] = -1 - a[q][k];
                }
            }
            if (__any(bad))
                break;
            // per tree of the column: points, largest contribution; (two rows per lane: both halves of a tree are taken together)
            unsigned long long todo[RPL];
#pragma unroll
            for (int k = 0; k < RPL; k++)
                todo[k] = __ballot(sl[k] >= 0);
            while (true)
            {
                int s0 = -1;
#pragma unroll
                for (int k = RPL - 1; k >= 0; k--)
                    if (todo[k])
                        s0 = __builtin_amdgcn_readlane(sl[k], __builtin_ctzll(todo[k]));
                if (s0 < 0)
                    break;
                int cnt = 0;
                unsigned long long mine = 0ull;
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const bool in = sl[k] == s0;
                    const unsigned long long mm = __ballot(in);
                    cnt += __popcll(mm);
                    todo[k] &= ~mm;
                    const unsigned long long fb = (unsigned long long) __double_as_longlong(pf_fin[q][k]);
                    if (in && fb > mine) // (non-negative doubles order like their bit patterns)
                        mine = fb;
                }
                const unsigned long long mx = wave_max_f64_bits(mine);
                unsigned long long m = __ballot(lane >= cidx && __longlong_as_double((long long) mx) > mz);
                if (lane == 0)
                {
                    if (T.birth[s0] == cidx)
                        m |= (1ull << cidx) - 1ull; // not listed before its column: nothing to finish there
                    atomicOr(&T.g_alive[s0], m);
                    atomicMax(&T.g_fin[s0], mx);
                    atomicAdd(&T.g_pts[s0], (unsigned) cnt);
                    atomicMax(&T.g_last[s0], cidx);
                }
            }
            AB_PHW(3)
            // link candidates (accepted candidates after the first, cc.cpp:693-694) that lead to another tree
            if ((lds_ld(&T.col_info[cidx]) >> 8) & 3) // (wave-uniform: the column has points with links at all)
            {
                T.pair[w][lane] = 0ull;
                wave_lds_fence();
            }
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int n = pf_nl[q][k] == 255 ? 0 : pf_nl[q][k];
                if (sl[k] >= 0 && n > 0)
                {
                    const long long gc = gc0 + cidx;
                    for (int j = 0; j < n; j++)
                    {
                        const int code = (int) ((pf_lk[q][k] >> (16 * j)) & 0xffff);
                        const int v = ring[(int) ((gc - (code >> 8)) & (AB_RING - 1)) * R + (code & 0xff)];
                        if (v >= 0)
                            bad = 1; // (cannot happen: every cell of the group is resolved)
                        else if (v > AB_NONE && -1 - v != sl[k])
                        {
                            const unsigned long long bit = 1ull << (-1 - v);
                            if (!(atomicOr(&T.pair[w][sl[k]], bit) & bit))
                            {
                                const int e = atomicAdd(&T.n_ev, 1);
                                if (e < AB_EVENTS)
                                    T.ev[e] = ((unsigned) cidx << 16) | ((unsigned) sl[k] << 8) | (unsigned) (-1 - v);
                            }
                        }
                    }
                }
            }
            AB_PHW(4)
        }
            if (__any(bad) && lane == 0)
                T.bail = AB_BAIL_DEAD;
            ab_barrier(); // B2
            AB_PHW(5)
            if (T.bail || T.n_ev > AB_EVENTS)
            {
                bailed = true;
                break;
            }
            {
                // the next group's inputs travel while the timeline wave works
                int l1 = lc0 + ncols;
                l1 = l1 >= RC ? l1 - RC : l1;
                prefetch_inputs(gc0 + ncols, l1);
            }
            ab_barrier(); // B3
            AB_PHW(6)
            if (T.bail)
            {
                bailed = true;
                break;
            }
            // ================================================================================== 4b: tree roots of the group's cells, slot ring
        const bool any_finished = T.any_finished != 0;
#pragma unroll
        for (int q = 0; q < AB_CPW; q++)
        {
            const int cidx = w + q * AB_WAVES;
            if (cidx < ncols)
            {
                int lc = lc0 + cidx;
                lc = lc >= RC ? lc - RC : lc;
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    if (row < R)
                    {
                        const int v = a[q][k];
                        p.root[lc * R + row] = v > AB_NONE ? T.cell_old[-1 - v] : -1;
                    }
                }
                const int first_local = T.col_fix[cidx];
                if (first_local >= 0)
                {
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                    {
                        const int row = k * 64 + lane;
                        if (row < R && a[q][k] > AB_NONE)
                        {
                            const int ci = lc * R + row;
                            const float mad = ccm::asinf_exact(cfg.max_distance / p.dist[ci]);
                            int dummy_root = -1, dummy_parent = -1, dummy_n = 0, vis = 0;
                            bool dummy_ov = false;
                            scan_point<false, false, true>(c, lc, gc0 + cidx, row, first_local, mad, 0., dummy_root, dummy_parent, nullptr, dummy_n, dummy_ov,
                                                           0, &vis);
                            p.sc_visits[ci] = sat_u16(vis);
                        }
                    }
                }
            }
        }
        if (any_finished)
        {
            // the look-back window of the next group: slots renumbered, finished trees marked
            const long long nb = gc0 + ncols;
            for (int i = threadIdx.x - 64; i < WIN_COLS * R; i += AB_THREADS - 64)
            {
                const int back = i / R + 1, row = i - (back - 1) * R;
                const int ri = (int) ((nb - back) & (AB_RING - 1)) * R + row;
                const int v = ring[ri];
                if (v > AB_NONE)
                {
                    const int r = T.remap[-1 - v];
                    ring[ri] = (short) (r >= 0 ? -1 - r : AB_DEAD);
                }
            }
        }
            // (no barrier here: the next group's first phase only writes ring columns, tree slots and words that nothing above reads)
            AB_PHW(8)
#if defined(CC_AB_STATS) && defined(CC_AB_STATS_W)
            ab_t[7]++;
#endif
            n_unf = s_nunf;
            gc0 += ncols;
            lc0 += ncols;
            lc0 = lc0 >= RC ? lc0 - RC : lc0;
            if (T.next_bail)
                bailed = true;
        }
    }

    // ---- persist the tree state back to the global planes (the serial kernels and the next batch load it from there) ------------------
    __syncthreads();
    n_unf = s_nunf;
    if ((int) threadIdx.x < n_unf)
    {
        const int i = threadIdx.x;
        const int cell = T.cell[i];
        p.ulist[i] = cell;
        p.t_pos[cell] = i;
        p.t_fin[cell] = __longlong_as_double((long long) T.fin[i]);
        p.t_width[cell] = (unsigned) (T.last[i] - T.gcol[i]) + 1u;
        p.t_pts[cell] = T.pts[i];
        p.t_uf[cell] = T.cell[T.comp[i]];
        p.t_cid[cell] = 0;
        p.t_finished[cell] = 0;
    }
    if (wave == 0)
    {
        double lb = lane < n_unf ? __longlong_as_double((long long) T.fin[lane]) : 1.7976931348623157e308;
        lb = wave_min_f64(lb);
        if (lane == 0)
        {
            st->first_unpublished = first_unpub;
            st->batch[slot].pub_end = first_unpub;
            st->ring_start = ring_start;
            st->cluster_counter = cluster_counter;
            st->n_unfinished = n_unf;
            if (n_unf > 0)
                st->min_required = T.gcol[0];
            st->finish_lower_bound = lb; // (a lower bound of every cluster's largest finished_at: the serial kernels only filter with it)
            st->last_round_min_az = last_min_az;
            st->cells_published = cells_published;
            st->clusters_finished = clusters_finished;
            st->stamp_alias_rounds = alias_rounds;
            st->batch[slot].acp_next = gc0;
            // a limited launch of the serial kernel takes the group that could not be taken here (and no more), then this kernel is tried again
            st->serial_until = bailed ? gc0 + AB_G : 0;
            st->batch_columns += batch_cols;
            st->batch_bails += bailed ? 1ull : 0ull;
            if (bailed)
            {
                st->batch_bail_reason[T.bail & 7] += 1ull;
                if (bail_count)
                    atomicAdd(bail_count, 1); // (the engine launches more (batch-parallel, serial) rounds for the next batches: assoc_rounds 0)
            }
            st->n_events = n_events < g.event_capacity ? n_events : g.event_capacity;
            if (g.record_events && n_events > g.event_capacity)
                raise_error(st, CC_ERR_CAPACITY, n_events, 0);
#if defined(CC_AB_STATS) && !defined(CC_AB_STATS_W)
            ab_t[9] = __builtin_amdgcn_s_memtime() - ab_t0;
            for (int i = 0; i < 10; i++)
                st->dbg[i] += ab_t[i];
#endif
        }
    }
#if defined(CC_AB_STATS) && defined(CC_AB_STATS_W)
    if (wave == 1 && lane == 0)
    {
        ab_t[9] = __builtin_amdgcn_s_memtime() - ab_t0;
        for (int i = 0; i < 10; i++)
            st->dbg[i] += ab_t[i];
    }
#endif
}
