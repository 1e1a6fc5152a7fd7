// Batch-parallel association + finished-cluster check (included by cc_kernels.h inside namespace cck, after cc_assoc3.h).
//
// k_assocb does what k_assoc3 / k_assoc_lds do (association bookkeeping cc.cpp:643-696 and 773-835, finished-cluster check :837-974, publish
// bookkeeping :1035-1092), but not as a walk over the columns: a block of AB_WAVES worker wavefronts + one timeline wavefront takes a GROUP of
// up to 60 columns of one stream at a time and nothing in it is serial per column.
//
//   1  tree of every point   k_scan left, per point, where its chain of same-column parents ends (sc_term): a new root of the column or a cell
//                            of an earlier column. Inside the group that is a forest of pointers, resolved by pointer jumping in an LDS ring
//                            of per-cell tree slots (<= 7 rounds for 64 columns, instead of one dependent look-up per column).
//   2  records               per column and RUN of rows that lead to one tree: points and largest finished_at contribution (cc.cpp:666-670),
//                            by one segmented DPP scan per column (no loop over the column's trees). What the finished-cluster check needs of
//                            them is ONE 64-bit word per tree: bit j = "this tree alone keeps its cluster unfinished at column j of the group"
//                            (its largest finished_at so far > the column's smallest azimuth, :884-885). A record (column c, value f)
//                            contributes ballot(lane >= c && f > min_az[lane]) — one compare over the lanes, one LDS atomic OR.
//   3  timeline (one wave)   lanes = trees. A cluster is unfinished at column j iff the OR of its trees' words has bit j set; it is finished at
//                            the first eligible column where it has not. Tree links (cc.cpp:675-696) are the only thing that changes clusters:
//                            the few links of a group that join two different trees are applied in column order, the columns between two such
//                            events are one "epoch" evaluated with a single segmented OR. Ids by rank of (finish column, oldest tree),
//                            first-unpublished column per column as the oldest tree still listed (cc.cpp:944-959), events by prefix sums.
//   4  commit                tree roots of the group's cells, finished trees, compaction of the tree table, remap of the slot ring.
//
// Round 5: the groups are PIPELINED. While the workers resolve, jump and record group i, the timeline wavefront evaluates and commits group
// i - 1 (round 4: it idled 40 % of a group waiting for the records, the workers 20 % waiting for the timeline). What makes that possible is a
// slot numbering that does not wait for the previous group's compaction: the workers number the trees of group i as
//     N(i) = [ survivors of the timeline pass of group i - 2, in list order ] ++ [ trees born in group i - 1 ] ++ [ trees born in group i ]
// — the list the timeline pass of group i - 1 works on, plus the group's own new roots (how many every column starts is static: k_scan's column
// summary). The timeline pass of group i translates: a survivor's slot is its position BEFORE the compaction of pass i - 1 (`sig`), a tree that
// pass i - 1 finished has no lane any more; a point or link of group i that names such a tree is what the serial algorithm refuses (cc.cpp:658,
// :688-690): points stop the kernel in front of the group (AB_BAIL_DEAD, as before), links are dropped (as before). The slot ring is renumbered
// by the workers one iteration late (compaction of pass i - 2 before group i's pointers are written), the tree roots of group i's cells are
// written when pass i has committed (one iteration after the workers recorded them; `a1` / `a2` keep two groups of resolved slots in registers).
// Per-group shared state is double-buffered by group parity (AbGroup: what the workers produce; AbHeader: what the timeline plans).
//
// One iteration (group i for the workers, i - 1 for the timeline), two block barriers + one per pointer-jumping round:
//     workers   points of group i requested, roots of group i - 2, ring renumbering, pointers of i | B1 | jumping (barrier per round) | records, links of i | B2
//     timeline  first half of the pass of group i - 1 (up to the finished clusters' ids)          | B1 | (follows the rounds)        | second half, header of i + 1 | B2
// (what a pass hands to the workers — compaction map, root cells, re-scan columns — is double-buffered too: the workers read the pass of group
// i - 2 while the pass of group i - 1 writes)
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
#ifndef CC_AB_MIN_WAVES_PER_SIMD
#define CC_AB_MIN_WAVES_PER_SIMD 1
#endif
constexpr int AB_WAVES = CC_AB_WAVES;      // worker wavefronts; one more wavefront runs the timeline
constexpr int AB_THREADS = 64 * (AB_WAVES + 1);
constexpr int AB_SUB = 4;                  // 64-row pieces of a worker wavefront's tile: 4 columns of a 64-row sensor, 2 columns of a 128-row sensor
constexpr int AB_G_MAX = AB_WAVES * AB_SUB; // columns per group at 64 rows (<= 64 = lanes of the timeline wave); AB_WAVES * AB_SUB / RPL in general
constexpr int AB_POINTS = AB_WAVES * 64;   // active points per group: one lane each
#ifndef CC_AB_HOPS
#define CC_AB_HOPS 2
#endif
constexpr int AB_HOPS = CC_AB_HOPS;        // pointer hops per jumping round
constexpr int AB_RING = 128;               // columns of the slot ring at 64 rows (power of two >= columns per group + WIN_COLS); half of it at 128 rows
constexpr int AB_TREES = 64;               // trees a group can see (unfinished two groups ago + born since) = lanes of the timeline wave
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
static_assert(AB_G_MAX <= 64 && AB_THREADS <= 1024 && AB_RING >= AB_G_MAX + WIN_COLS, "group geometry");
static_assert(LINK_SLOTS == 4, "the link phase reads the four candidate codes of a point at once");

// what the workers produce for a group (indexed by the slots N(i)); read by the timeline one iteration later. Two buffers (group parity)
struct AbGroup
{
    unsigned long long g_alive[AB_TREES]; // bit j: the group's records alone keep the tree's cluster unfinished at column j of the group
    unsigned long long g_fin[AB_TREES];   // largest finished_at contribution of the group (bits of a non-negative double)
    unsigned g_pts[AB_TREES];
    int g_last[AB_TREES];                 // last column of the group (relative) that attached a point, -1 none
    int birth[AB_TREES];                  // column of the group (relative) the tree starts in, -1: older
    int g_cell[AB_TREES];                 // root cell of the trees born in the group
    unsigned ev[AB_EVENTS];               // column << 16 | tree a << 8 | tree b
    int n_ev;
    int wbail;                            // the workers cannot take the group (nothing of it is committed)
    int jflag[8];                         // pointer jumping: round r left pointers unresolved
};
// what the timeline plans for a group. Two buffers (group parity)
struct AbHeader
{
    int col_base[64]; // new roots of the group's earlier columns
    int col_info[64];
    int ppos[64];     // active points of the group's earlier columns = where the column's points go in the packed order
    int chunk_col[16]; // the column the first point of wavefront w belongs to
    unsigned char colstart[AB_POINTS]; // column + 1 at the position of the column's first point, 0 elsewhere
    double min_az[64];
    long long gc0;    // first global column of the group
    int lc0;
    int ncols;
    int nborn;
    int tot;          // active points of the group
    int base;         // slots in front of the group's own new roots = size of the list the previous group's timeline pass works on
    int bail;         // the kernel stops in front of this group
};
struct AbTrees
{
    // persistent over the groups: the unfinished trees in creation order (the reference's sc_unfinished_point_trees_); timeline only
    int cell[AB_TREES];
    long long gcol[AB_TREES];
    unsigned long long fin[AB_TREES]; // bits of finished_at_continuous_azimuth_angle (non-negative double)
    long long last[AB_TREES];         // last global column that attached a point
    unsigned pts[AB_TREES];
    int comp[AB_TREES];               // cluster = smallest list position of its trees
    int sig[AB_TREES];                // the tree's slot in the numbering of the group the next pass evaluates (its position before the last compaction)
    // scratch of a timeline pass
    unsigned long long k_or[AB_TREES];
    unsigned k_pts[AB_TREES];
    long long k_max[AB_TREES];
    unsigned k_cid[AB_TREES];
    long long t_gcol[AB_TREES];
    int t_birth[AB_TREES];
    int lane_of[AB_TREES];            // slot -> lane of the pass, -1: the tree finished in the pass before
    int col_ncl[64];
    int col_ebase[64];
    // timeline pass -> workers: the pass of group j writes buffer j & 1 while the workers still read what the pass of group j - 1 left
    int remap[2][AB_TREES];           // list position of the pass -> position after its compaction, -1 finished
    int cell_by_slot[2][AB_TREES];    // slot (numbering of the pass's group) -> root cell
    int col_fix[2][64];               // >= 0: ring column of the first unpublished column while the column was associated (visit counts are re-taken)
    int rm_n[2];                      // size of the list the pass worked on
    int rm_nunf[2];                   // trees it left
    int any_finished[2];
    int tbail;                        // the pass could not take its group (nothing of it is committed)
    AbGroup G[2];
    AbHeader H[2];
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
// inclusive prefix maximum of non-negative values
__device__ __forceinline__ int wave_incl_max_i32(int v)
{
    int t;
    t = dpp_mov_i32<0x111, 0xf>(0, v); v = t > v ? t : v;
    t = dpp_mov_i32<0x112, 0xf>(0, v); v = t > v ? t : v;
    t = dpp_mov_i32<0x114, 0xf>(0, v); v = t > v ? t : v;
    t = dpp_mov_i32<0x118, 0xf>(0, v); v = t > v ? t : v;
    t = dpp_mov_i32<0x142, 0xa>(0, v); v = t > v ? t : v;
    t = dpp_mov_i32<0x143, 0xc>(0, v); v = t > v ? t : v;
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
// Maximum over the RUN of lanes [head, lane] a lane belongs to (`head` = first lane of the run, the same in all its lanes): the values are bit
// patterns of non-negative doubles (finished_at), a source lane outside the run contributes + 0.0. One step = two DPP moves, one compare, two
// selects, one v_max_f64; six steps cover the wavefront.
template<int CTRL, int ROW_MASK>
__device__ __forceinline__ long long ab_segmax_step(long long b, const int src_lane, const int head)
{
    long long t = dpp_mov_i64<CTRL, ROW_MASK>(0ll, b);
    t = src_lane >= head ? t : 0ll;
    return __double_as_longlong(__builtin_fmax(__longlong_as_double(t), __longlong_as_double(b)));
}
__device__ __forceinline__ unsigned long long ab_run_max_f64_bits(unsigned long long bits, const int lane, const int head)
{
    long long b = (long long) bits;
    b = ab_segmax_step<0x111, 0xf>(b, lane - 1, head);
    b = ab_segmax_step<0x112, 0xf>(b, lane - 2, head);
    b = ab_segmax_step<0x114, 0xf>(b, lane - 4, head);
    b = ab_segmax_step<0x118, 0xf>(b, lane - 8, head);
    b = ab_segmax_step<0x142, 0xa>(b, (lane & ~15) - 1, head); // (rows 1 and 3 take lane 15 / 47: everything of the row below)
    b = ab_segmax_step<0x143, 0xc>(b, 31, head);               // (rows 2 and 3 take lane 31: everything of rows 0 - 1)
    return (unsigned long long) b;
}

// the block's LDS (one object, declared by the kernel: k_small_all fills part of it before assocb_body runs)
template<int RPL>
struct AbShared
{
    AbTrees T;
    short ring[(AB_RING / RPL) * WAVE * RPL]; // tree slot of every cell of the last AB_RING / RPL columns
    int s_nunf;
    long long s_gc_final;
    int s_serial_cols;
    int s_reason;
    unsigned long long s_cols;
    int s_view_done; // (k_small_all / k_resident, phase F: wavefronts that have written their share of the call's column views)
};

// what assocb_body needs of the stream's state before its first group: the persistent tree state (global planes indexed by root cell) -> LDS, list
// position = slot, and the slot ring of the WIN_COLS columns before col_begin. Nothing of it depends on the batch that is being inserted, so
// k_small_all lets its idle wavefronts do this (threads tid of nthreads) next to the call's front half, with col_begin predicted.
template<int RPL>
__device__ __forceinline__ void assocb_load_state(const Geometry& g, const SP& p, AbShared<RPL>& S, const long long col_begin, const long long first_column,
                                                  const int n_unf, const int tid, const int nthreads)
{
    constexpr int RING = AB_RING / RPL;
    const int R = g.num_rows, RC = g.ring_cols;
    AbTrees& T = S.T;
    for (int i = tid; i < n_unf; i += nthreads)
    {
        const int cell = p.ulist[i];
        const long long tg = p.colg[cell / R];
        T.cell[i] = cell;
        T.gcol[i] = tg;
        T.fin[i] = (unsigned long long) __double_as_longlong(p.t_fin[cell]);
        T.last[i] = tg + (long long) p.t_width[cell] - 1;
        T.pts[i] = p.t_pts[cell];
        T.comp[i] = p.t_pos[p.t_uf[cell]];
        T.sig[i] = i;
    }
    // slot ring of the WIN_COLS columns before col_begin: two dependent gathers per cell (root plane, then the tree planes at the root)
    for (int i = tid; i < WIN_COLS * R; i += nthreads)
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
        S.ring[(int) (gcx & (RING - 1)) * R + row] = (short) v;
    }
}

// one wavefront, behind assocb_load_state (and a block barrier): union-find parents -> cluster representative = the smallest list position of the
// set (the serial kernels that may have left this state number their trees by ids from a free ring: the root of a set is its smallest ID, which
// need not be its oldest tree)
template<int RPL>
__device__ __forceinline__ void assocb_representatives(AbShared<RPL>& S, const int n_unf)
{
    AbTrees& T = S.T;
    const int lane = lane_id();
    int rep = lane < n_unf ? T.comp[lane] : 0;
    for (int it = 0; it < 6; it++)
    {
        const int r2 = __shfl(rep, rep);
        rep = r2;
    }
    T.remap[0][lane] = AB_TREES;
    wave_lds_fence();
    if (lane < n_unf)
        atomicMin(&T.remap[0][rep], lane);
    wave_lds_fence();
    if (lane < n_unf)
        T.comp[lane] = lds_ld(&T.remap[0][rep]);
    if (lane == 0)
    {
        T.tbail = 0;
        T.any_finished[0] = T.any_finished[1] = 0;
    }
}

// what k_small_all's idle wavefronts prepared of the above, and for which state: assocb_body uses it when the prediction held
struct AbPreloaded
{
    int valid;
    long long col_begin, first_column;
    int n_unf;
};

// (a device function: k_assocb is its kernel; k_small_all runs it behind the front half of a small call, in the same block of AB_THREADS threads.
// Every early return is taken by the whole block.)
template<int RPL>
__device__ __forceinline__ void assocb_body(const Geometry& g, const cc_config& cfg, const Planes& P, StreamState* states, const int s, const int slot,
                                            int* __restrict__ bail_count, AbShared<RPL>& S, const AbPreloaded pre)
{
    constexpr int GCOLS = AB_WAVES * AB_SUB / RPL; // columns per group
#ifdef CC_AB_STATS
    const unsigned long long ab_entry = __builtin_amdgcn_s_memtime();
    const unsigned long long ab_entry_rt = wall_clock64(); // (constant 100 MHz: what s_memtime ticks at, measured)
#endif
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
    AbTrees& T = S.T;
    constexpr int RING = AB_RING / RPL; // columns of the slot ring
    static_assert(RING >= GCOLS + WIN_COLS, "slot ring");
    short* const ring = S.ring;
    int& s_nunf = S.s_nunf;
    long long& s_gc_final = S.s_gc_final;
    int& s_serial_cols = S.s_serial_cols;
    int& s_reason = S.s_reason;
    unsigned long long& s_cols = S.s_cols;

    const long long col_begin = st->batch[slot].acp_next, col_end = st->batch[slot].seg_end, first_column = st->first_column;
    long long first_unpub = st->first_unpublished, ring_start = st->ring_start;
    unsigned long long cluster_counter = st->cluster_counter;
    double last_min_az = st->last_round_min_az;
    unsigned long long cells_published = st->cells_published, clusters_finished = st->clusters_finished;
    unsigned long long alias_rounds = st->stamp_alias_rounds;
    int n_events = st->n_events;

    // ---- the persistent tree state -> LDS, the slot ring of the columns in front of the batch (unless the caller's idle wavefronts did it) -------
    const bool preloaded = pre.valid != 0 && pre.col_begin == col_begin && pre.first_column == first_column && pre.n_unf == n_unf;
    if (!preloaded)
    {
        assocb_load_state<RPL>(g, p, S, col_begin, first_column, n_unf, (int) threadIdx.x, AB_THREADS);
        __syncthreads();
        if (wave == 0)
            assocb_representatives<RPL>(S, n_unf);
    }
    if (wave == 0 && lane == 0 && st->batch[slot].pub_begin < 0)
        st->batch[slot].pub_begin = first_unpub; // first association kernel of this pass
    __syncthreads();

#ifdef CC_AB_STATS
    unsigned long long ab_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ab_mark = __builtin_amdgcn_s_memtime();
    const unsigned long long ab_t0 = ab_mark;
    ab_t[13] = ab_mark - ab_entry; // prologue
#define AB_PH_(i)                                                    \
    {                                                                \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
        ab_t[i] += n_ - ab_mark;                                     \
        ab_mark = n_;                                                \
    }
#else
#define AB_PH_(i)
#endif
    // CC_AB_STATS: the phase clocks of the timeline wavefront; with CC_AB_STATS_W those of worker wavefront 1 instead (tools/prof_assocb.py)
#if defined(CC_AB_STATS) && defined(CC_AB_STATS_W)
#define AB_PH(i)
#ifndef CC_AB_STATS_WAVE
#define CC_AB_STATS_WAVE 1
#endif
#define AB_PHW(i)                   \
    if (wave == CC_AB_STATS_WAVE) \
    AB_PH_(i)
#elif defined(CC_AB_STATS)
#define AB_PH(i) AB_PH_(i)
#define AB_PHW(i)
#else
#define AB_PH(i)
#define AB_PHW(i)
#endif
    const double mz_inf = 1.7976931348623157e308;

    // Two roles, one barrier schedule per iteration (file header). The timeline wavefront and the workers run separate loops (separate register
    // allocations: the workers hold a group of prefetched inputs and two groups of resolved slots, the timeline wave ~60 per-tree / per-column
    // values) that meet at B1, the pointer-jumping rounds and B2. Both sides take the same decision behind B2 from the same LDS words:
    //   T.tbail                                              the pass of group i - 1 failed -> stop, nothing of groups >= i - 1 is committed
    //   group i missing / its header says stop / G.wbail     stop behind group i - 1 (the workers still write its roots)
    if (wave == 0)
    {
        // =============================================================================================== timeline wavefront
        // column summaries of the group after the next header, requested a whole iteration ahead (registers, lanes = columns)
        int nx_info = 0, nx_act = 0;
        double nx_maz = mz_inf;
        auto load_next_info = [&](const long long g0, const int l0)
        {
            nx_info = 0;
            nx_act = 0;
            nx_maz = mz_inf;
            if (lane < GCOLS && g0 + lane < col_end)
            {
                int lcj = l0 + lane;
                lcj = lcj >= RC ? lcj - RC : lcj;
                nx_info = p.col_info[lcj];
                nx_act = p.col_act[lcj];
                nx_maz = p.colminaz[lcj];
            }
        };
        // group header: how many columns the group takes (the trees it may see must fit the lanes), what stops the kernel in front of it; the
        // group's AbGroup buffer is wiped; then the summaries of the group after it are requested.
        // `base` = trees the workers number in front of the group's own new roots; `oldest` = a lower bound of the column of every tree among them
        auto group_header = [&](const int b, const long long g0, const int l0, const int base, const long long oldest)
        {
            AbHeader& H = T.H[b];
            AbGroup& G = T.G[b];
            const int info = nx_info;
            const double maz = nx_maz;
            const bool valid = lane < GCOLS && g0 + lane < col_end;
            const int cnt = info & 0xff;
            const int incl = wave_incl_add_i32(cnt);
            const int pts = valid ? (nx_act & 0xff) + (nx_act >> 8) : 0;
            const int ipts = wave_incl_add_i32(pts);
            // (the group's points must fit the lanes of the worker wavefronts; one column always does.) New roots: a group takes at most HALF of the
            // tree lanes `base` leaves (its first column aside) — the next group's `base` still contains all of this group's new roots, so a group that
            // fills the lanes leaves the next one no room for a single column and the kernel stops (AB_BAIL_TREES: what vegetation-like streams did
            // once per rotation); with the half rule the room shrinks geometrically and the finished trees of the passes in between give it back
            const int born_cap = (tree_limit - base + 1) >> 1;
            const unsigned long long okm = __ballot(valid && base + incl <= tree_limit && ipts <= AB_POINTS && (lane == 0 || incl <= born_cap));
            const int ncols = ~okm ? __builtin_ctzll(~okm) : 64;
            const unsigned long long cm = ncols >= 64 ? ~0ull : ((1ull << ncols) - 1ull);
            int bail = (ncols == 0 && g0 < col_end) ? AB_BAIL_TREES : 0;
            if (__ballot(((info >> 8) & 1) != 0) & cm)
                bail = AB_BAIL_LINKS; // a point with more link candidates than k_scan records (cc.cpp:693-694 would see them all)
            // no tree or cluster of this group can reach the one-rotation limits (cc.cpp:657, 913-924) while the oldest tree that may still be
            // unfinished is less than a rotation behind the group's last column
            if (base > 0 && ncols > 0 && (g0 + ncols - oldest) >= NC)
                bail = AB_BAIL_ROTATION;
            H.col_base[lane] = incl - cnt;
            H.col_info[lane] = info;
            H.ppos[lane] = ipts - pts;
            // where the columns begin in the packed order: markers for the workers' prefix maximum, and the column every wavefront starts in
            if (lane < AB_POINTS / 16)
                *(uint4*) &H.colstart[16 * lane] = make_uint4(0u, 0u, 0u, 0u);
            wave_lds_fence();
            if (lane < ncols && pts > 0)
            {
                H.colstart[ipts - pts] = (unsigned char) (lane + 1);
                for (int cw = (ipts - pts + 63) >> 6; 64 * cw < ipts && cw < 16; cw++)
                    H.chunk_col[cw] = lane;
            }
            H.min_az[lane] = maz;
            G.g_alive[lane] = 0ull;
            G.g_fin[lane] = 0ull;
            G.g_pts[lane] = 0u;
            G.g_last[lane] = -1;
            G.birth[lane] = -1;
            const int nb = __builtin_amdgcn_readlane(incl, ncols > 0 ? ncols - 1 : 0);
            const int npts = __builtin_amdgcn_readlane(ipts, ncols > 0 ? ncols - 1 : 0);
            if (lane < 8)
                G.jflag[lane] = 0;
            if (lane == 0)
            {
                G.n_ev = 0;
                G.wbail = 0;
                H.gc0 = g0;
                H.lc0 = l0;
                H.ncols = bail ? 0 : ncols;
                H.nborn = (ncols > 0 && !bail) ? nb : 0;
                H.tot = (ncols > 0 && !bail) ? npts : 0;
                H.base = base;
                H.bail = bail;
            }
            int l1 = l0 + ncols;
            l1 = l1 >= RC ? l1 - RC : l1;
            load_next_info(g0 + ncols, l1);
        };
        load_next_info(col_begin, (int) (col_begin % RC));
        group_header(0, col_begin, (int) (col_begin % RC), n_unf, n_unf > 0 ? T.gcol[0] : col_begin);
        wave_lds_fence();
        ab_barrier(); // P0: header of group 0

        // the group the next pass evaluates: its first column, kept from the header
        long long tg0 = col_begin;
        int tlc0 = (int) (col_begin % RC);
        unsigned long long batch_cols = 0;
        long long gc_final = col_begin;
        int serial_cols = GCOLS; // columns the serial kernel takes behind a stop: the group that could not be taken
        int reason = 0;
        // the barriers in the middle of an iteration: the workers' pointers are written (B1), one per jumping round
        auto mid_barriers = [&](const int b)
        {
            ab_barrier(); // B1
            AB_PH(1)
            for (int r = 0; r < 8; r++)
            {
                ab_barrier();
                if (!T.G[b].jflag[r])
                    break;
            }
            AB_PH(2)
        };
        for (int i = 0;; i++)
        {
            const int b = i & 1, pb = b ^ 1;
            const bool have_prev = i > 0; // (a group i - 1 exists whenever the loop got here)
            // ---- the own alive words of the pass's old trees (what a tree that was listed before the group contributes to the check of
            //      the group's columns by the finished_at it came with): lanes = columns for the ballots, lane t keeps tree t's word
            unsigned long long alive_old = 0ull;
            if (have_prev)
            {
                const double mzp = T.H[pb].min_az[lane];
                const double f_mine = lane < n_unf ? __longlong_as_double((long long) T.fin[lane]) : 0.;
                for (int t = 0; t < n_unf; t++)
                {
                    const double f = lane_f64(f_mine, t);
                    const unsigned long long m = __ballot(f > mzp);
                    alive_old = lane == t ? m : alive_old;
                }
            }
            AB_PH(0)
            // ================================================================================== 3: timeline pass of group i - 1
            if (have_prev)
            {
                const AbHeader& H = T.H[pb];
                AbGroup& G = T.G[pb];
                const long long gc0 = tg0;
                const int lc0 = tlc0;
                const int ncols = H.ncols;
                const int nborn = H.nborn;
                const int base = H.base;
                const double mz = H.min_az[lane]; // lanes = columns of the group
                const int n = n_unf + nborn;
                const bool is_t = lane < n;
                const bool old = lane < n_unf;
                const int sig = old ? T.sig[lane] : base + (lane - n_unf); // the tree's slot in the workers' numbering of this group
                T.lane_of[lane] = -1;
                wave_lds_fence();
                if (is_t)
                    T.lane_of[sig] = lane;
                wave_lds_fence();
                // a tree the pass before finished that received a point of this group: the reference refuses the attach (cc.cpp:658)
                int tbad = (lane < base && lds_ld(&T.lane_of[lane]) < 0 && G.g_last[lane] >= 0) ? AB_BAIL_DEAD : 0;
                const int birth = is_t ? G.birth[sig] : -1;
                const int cell = old ? T.cell[lane] : (is_t ? G.g_cell[sig] : 0);
                const long long tg = old ? T.gcol[lane] : gc0 + birth;
                // (a tree is not listed before its column: nothing to finish there)
                const unsigned long long alive_w = is_t ? (G.g_alive[sig] | alive_old | (birth > 0 ? (1ull << birth) - 1ull : 0ull)) : ~0ull;
                const int g_last = is_t ? G.g_last[sig] : -1;
                int comp = old ? T.comp[lane] : lane;
                int fc = 64; // column of the group (relative) whose check finishes the tree's cluster; 64 = not in this group
                T.t_gcol[lane] = tg;
                T.t_birth[lane] = birth;
                T.col_ncl[lane] = 0;
                // rounds that can finish something: not the ones whose smallest azimuth equals the previous round's (the BFS of cc.cpp:854 then
                // meets its own visited stamp everywhere)
                const double prev_az = wave_shr1_f64(mz);
                const bool alias = lane < ncols && mz == (lane == 0 ? last_min_az : prev_az);
                const unsigned long long colmask = ncols >= 64 ? ~0ull : ((1ull << ncols) - 1ull);
                const unsigned long long alias_m = __ballot(alias);
                const unsigned long long elig = ~alias_m & colmask;
                const int nev_raw = G.n_ev;
                const int nev = nev_raw < AB_EVENTS ? nev_raw : AB_EVENTS;
                const unsigned evw = lane < nev ? G.ev[lane] : 0u;
                const int evcol = (int) (evw >> 16);
                // (slots -> lanes of this pass; a link to a tree the pass before finished is refused like any link to a finished tree, cc.cpp:688-690)
                const int eva = lane < nev ? lds_ld(&T.lane_of[(evw >> 8) & 0xff]) : -1, evb = lane < nev ? lds_ld(&T.lane_of[evw & 0xff]) : -1;
                bool ev_done = lane >= nev || eva < 0 || evb < 0;
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
                if (is_t && fc < 64 && g_last > fc)
                    tbad = AB_BAIL_LATE;

                // ---- finished clusters: points, extent, ids in the order the reference's BFS meets them (column, then oldest tree)
                const unsigned pts_t = (old ? T.pts[lane] : 0u) + (is_t ? G.g_pts[sig] : 0u);
                const long long last_old = old ? T.last[lane] : -1;
                const long long last_t = (g_last >= 0 && gc0 + g_last > last_old) ? gc0 + g_last : last_old;
                const unsigned long long fin_old = old ? T.fin[lane] : 0ull;
                const unsigned long long fin_g = is_t ? G.g_fin[sig] : 0ull;
                const unsigned long long fin_t = fin_g > fin_old ? fin_g : fin_old;
                const bool fin_here = is_t && fc < 64;
                const bool any_fin = __any(fin_here);
                unsigned cpts = 0u, cid = 0u, cid_t = 0u;
                long long cmax = -1;
                int rank_col = 0, n_ids = 0;
                bool has_id = false;
                if (any_fin) // (wave-uniform: most groups finish nothing)
                {
                    T.k_pts[lane] = 0u;
                    T.k_max[lane] = -1;
                    T.k_cid[lane] = 0u;
                    wave_lds_fence();
                    if (fin_here)
                    {
                        atomicAdd(&T.k_pts[comp], pts_t);
                        atomicMax(&T.k_max[comp], last_t);
                    }
                    wave_lds_fence();
                    const bool is_rep = fin_here && comp == lane;
                    cpts = lds_ld(&T.k_pts[lane]);
                    cmax = lds_ld(&T.k_max[lane]);
                    has_id = is_rep && cpts > 5u; // cc.cpp:936
                    const int key = has_id ? fc * 64 + lane : 0x7fffffff;
                    int rank = 0;
                    for (int q = 0; q < n; q++)
                    {
                        const int ki = __builtin_amdgcn_readlane(key, q);
                        rank += ki < key ? 1 : 0;
                        rank_col += (ki < key && (ki >> 6) == fc) ? 1 : 0;
                    }
                    cid = (unsigned) (cluster_counter + (unsigned long long) rank);
                    n_ids = __popcll(__ballot(has_id));
                    if (has_id)
                    {
                        T.k_cid[lane] = cid;
                        atomicAdd(&T.col_ncl[fc], 1);
                    }
                    wave_lds_fence();
                    cid_t = fin_here ? lds_ld(&T.k_cid[comp]) : 0u;
                }

                AB_PH(4)
                // ---- the second half of the pass runs next to the workers' records and links (the first half next to their packing and pointers)
                mid_barriers(b);
                // ---- per column (lanes = columns): the oldest tree still listed when the column's check ends (cc.cpp:944-959)
                int idx = -1;
                for (int q = n - 1; q >= 0; q--)
                {
                    const int fi = __builtin_amdgcn_readlane(fc, q);
                    if (fi >= lane)
                        idx = q;
                }
                const long long gcj = gc0 + lane;
                long long mc = gcj + 1;
                bool listed = false;
                if (idx >= 0)
                {
                    const int bi = T.t_birth[idx];
                    if (bi <= lane)
                    {
                        mc = T.t_gcol[idx];
                        listed = true;
                    }
                }
                const long long mc_prev = wave_shr1_i64(mc);
                const long long fu = lane == 0 ? first_unpub : mc_prev; // first unpublished column while column j is associated
                const int info = H.col_info[lane];
                const int reach = (info >> 16) & 0xff;
                if (lane < ncols && (mc < fu || gcj - reach < fu) && tbad != AB_BAIL_DEAD)
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
                const int ncl = T.col_ncl[lane];
                const int per_col = lane < ncols ? 2 + ncl : 0;
                const int eincl = wave_incl_add_i32(per_col);
                T.col_ebase[lane] = n_events + eincl - per_col;
                wave_lds_fence();

                if (__any(tbad) || nev_raw > AB_EVENTS)
                {
                    // (an attach to a finished tree first, then a late point, then the reach)
                    int why = nev_raw > AB_EVENTS ? AB_BAIL_LINKS : AB_BAIL_REACH;
                    why = __any(tbad == AB_BAIL_LATE) ? AB_BAIL_LATE : why;
                    why = __any(tbad == AB_BAIL_DEAD) ? AB_BAIL_DEAD : why;
                    if (lane == 0)
                        T.tbail = why;
                }
                else
                {
                    // ============================================================================== 4a: commit (still wave 0)
                    T.col_fix[pb][lane] = fix; // cc.cpp:762-763: the live scan would have stopped earlier (or the bookkeeping error of :1072-1075: the serial kernel reports it)
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
                        const int cell_a = __shfl(cell, eva < 0 ? 0 : eva), cell_b = __shfl(cell, evb < 0 ? 0 : evb);
                        if (ev_made)
                            log_link(g, st, p.link_log, cell_a, cell_b);
                    }
                    // stable compaction of the list
                    const bool surv = is_t && fc == 64;
                    const unsigned long long sm = __ballot(surv);
                    const int np = __popcll(sm & lanes_below());
                    T.remap[pb][lane] = surv ? np : -1;
                    if (is_t)
                        T.cell_by_slot[pb][sig] = cell;
                    wave_lds_fence();
                    if (surv)
                    {
                        T.cell[np] = cell;
                        T.gcol[np] = tg;
                        T.fin[np] = fin_t;
                        T.last[np] = last_t;
                        T.pts[np] = pts_t;
                        T.comp[np] = lds_ld(&T.remap[pb][comp]);
                        T.sig[np] = lane;
                    }
                    const int n_left = __popcll(sm);
                    if (lane == 0)
                    {
                        T.any_finished[pb] = n_left != n ? 1 : 0;
                        T.rm_n[pb] = n;
                        T.rm_nunf[pb] = n_left;
                    }
                    n_unf = n_left;
                    batch_cols += (unsigned long long) ncols;
                    wave_lds_fence();
                }
            }
            else
                mid_barriers(b);
            AB_PH(6)
            // ---- header of group i + 1 (the pass above has read everything of the buffers it reuses). `base`: what the pass left + the new roots of group i
            const bool t_failed = have_prev && lds_ld(&T.tbail) != 0;
            if (!t_failed)
            {
                const AbHeader& Hi = T.H[b];
                const int ncols_i = Hi.ncols;
                const long long gi0 = Hi.gc0;
                const int li0 = Hi.lc0;
                const int nborn_i = Hi.nborn;
                int l1 = li0 + ncols_i;
                l1 = l1 >= RC ? l1 - RC : l1;
                const long long oldest = n_unf > 0 ? lds_ld(&T.gcol[0]) : gi0;
                if (ncols_i > 0)
                    group_header(pb, gi0 + ncols_i, l1, n_unf + nborn_i, oldest);
                // the next pass evaluates group i
                tg0 = gi0;
                tlc0 = li0;
            }
            wave_lds_fence();
            AB_PH(3)
            ab_barrier(); // B2
            AB_PH(5)
#ifdef CC_AB_STATS
            ab_t[7]++;
#endif
            // ---- the common decision
            if (t_failed)
            {
                gc_final = tg0; // (not advanced above: still the first column of group i - 1)
                reason = T.tbail;
                serial_cols = T.H[pb].ncols; // (its header is still in place: the next one was not written)
                break;
            }
            const int wb = T.G[b].wbail;
            if (T.H[b].ncols == 0 || wb)
            {
                gc_final = T.H[b].gc0;
                reason = wb ? wb : T.H[b].bail;
                serial_cols = wb ? T.H[b].ncols : GCOLS;
                break;
            }
        }
        if (lane == 0)
        {
            s_nunf = n_unf;
            s_gc_final = gc_final;
            s_serial_cols = serial_cols;
            s_reason = reason;
            s_cols = batch_cols;
        }
    }
    else
    {
        // =============================================================================================== worker wavefronts
#ifndef CC_AB_WPRIO
#define CC_AB_WPRIO 2
#endif
        __builtin_amdgcn_s_setprio(CC_AB_WPRIO); // (one below the timeline wavefront: it shares its SIMD with three workers)
        // Round 5: everything per POINT runs on the group's ACTIVE points packed into the lanes of the block. k_scan leaves every column's active
        // points packed in row order (pk_meta / pk_fin / pk_lk) and counts them (col_act); the header turns the counts into positions in the group's
        // (column, row) order, and wavefront w takes points 64 w .. 64 w + 63: pointers, jumping, records, links and tree roots run once per
        // wavefront on full lanes — a quarter to a third of the cells carries an obstacle point, so rows-as-lanes did the same work four times on
        // mostly idle lanes — with the same load on every wavefront whatever the scene puts into which column. The kernel is bound by the vector
        // issue of the ONE compute unit a stream's block sits on (4 wavefronts per SIMD; DPP, compare and 64-bit instructions at 4 clocks each):
        // instructions per group, not latency, set its time.
        const int w = wave - 1;
        unsigned long long* pair = T.pair[w];
        // the last two groups: the wavefront's points (where they lie, their resolved slots in the numbering of their own group) and where the groups
        // lie. The tree roots of a group's cells are written when the timeline pass of the group has committed, two iterations after the slots were
        // resolved (cells without a point got their -1 from k_scan)
        int loc1 = -1, loc2 = -1; // column of the group | row << 6, -1: no point in this lane
        int a1 = AB_NONE, a2 = AB_NONE;
        long long g1_gc0 = 0, g2_gc0 = 0;
        int g1_lc0 = 0, g2_lc0 = 0;
        // 4b of a committed group: tree roots of its points' cells, re-taken visit counts (mirror mode)
        auto write_roots = [&](const int tb, const int loc, const int av, const long long gc0, const int lc0)
        {
            if (loc >= 0)
            {
                const int cidx = loc & 63, row = loc >> 6;
                int lc = lc0 + cidx;
                lc = lc >= RC ? lc - RC : lc;
                const int ci = lc * R + row;
                at32(p.root, (unsigned) ci) = av > AB_NONE ? T.cell_by_slot[tb][-1 - av] : -1;
                if (g.mirror_fields && av > AB_NONE)
                {
                    const int first_local = T.col_fix[tb][cidx];
                    if (first_local >= 0)
                    {
                        const float mad = ccm::asinf_exact(cfg.max_distance / p.dist[ci]);
                        int dummy_root = -1, dummy_parent = -1, dummy_n = 0, vis = 0;
                        bool dummy_ov = false;
                        scan_point<false, false, true>(c, lc, gc0 + cidx, row, first_local, mad, 0., dummy_root, dummy_parent, nullptr, dummy_n, dummy_ov, 0, &vis);
                        p.sc_visits[ci] = sat_u16(vis);
                    }
                }
            }
        };
        ab_barrier(); // P0
        int why_stop = 0, last_tb = 0; // 1: the timeline pass failed (nothing owed), 2: stop behind group i - 1 (its roots are owed)
        for (int i = 0;; i++)
        {
            const int b = i & 1;
            const AbHeader& H = T.H[b];
            AbGroup& G = T.G[b];
            const int ncols = H.ncols;
            const long long gc0 = H.gc0;
            const int lc0 = H.lc0;
            const int base = H.base;
            // ================================================================================== 0: this wavefront's 64 points of the group: where they
            //      lie (a marker at the position of every column's first point, prefix maximum over the lanes), their packed records requested
            const bool act = 64 * w + lane < H.tot;
            int cidx = 0;
            unsigned meta = 0u;
            double pfin_d = 0.;
            unsigned long long plk = 0ull;
            {
                const int mk = act ? (int) H.colstart[64 * w + lane] : 0; // column + 1 where a column's points begin
                const int cs = wave_incl_max_i32(mk);
                cidx = cs > 0 ? cs - 1 : H.chunk_col[w];
                if (act)
                {
                    int lc = lc0 + cidx;
                    lc = lc >= RC ? lc - RC : lc;
                    const unsigned pi = (unsigned) (lc * R + (64 * w + lane - H.ppos[cidx]));
                    meta = at32(p.pk_meta, pi);
                    pfin_d = at32(p.pk_fin, pi);
                    if ((H.col_info[cidx] >> 8) & 3) // (the column has points with links; stale where this point has none: never looked at)
                        plk = at32(p.pk_lk, pi);
                }
            }
            AB_PHW(11)
            // ================================================================================== 4b of group i - 2 (its pass committed in iteration i - 1)
            if (i >= 2)
            {
                write_roots(b, loc2, a2, g2_gc0, g2_lc0);
                if (T.any_finished[b])
                {
                    // the look-back window of group i: slots renumbered by the compaction of the pass of group i - 2 (numbering N(i - 1) -> N(i)),
                    // finished trees marked (cells without a point hold stale values nobody follows)
                    const int rm_n = T.rm_n[b], shift = T.rm_nunf[b] - rm_n;
                    for (int j = threadIdx.x - 64; j < WIN_COLS * R; j += AB_THREADS - 64)
                    {
                        const int back = j / R + 1, row = j - (back - 1) * R;
                        const int rj = (int) ((gc0 - back) & (RING - 1)) * R + row;
                        const int v = ring[rj];
                        if (v > AB_NONE && v < 0)
                        {
                            const int sl = -1 - v;
                            const int r = sl < rm_n ? T.remap[b][sl & (AB_TREES - 1)] : sl + shift;
                            ring[rj] = (short) (r >= 0 ? -1 - r : AB_DEAD);
                        }
                    }
                    // (the window's columns are written by this loop only, group i's cells by the pointers below only: B1 in front of the rounds
                    // that read both orders them)
                }
            }
            AB_PHW(8)
            const double mz = H.min_az[lane]; // lanes = columns of the group
            // ================================================================================== 1: initial pointers
            const unsigned long long pfin = (unsigned long long) __double_as_longlong(pfin_d);
            int loc = -1;
            int a = AB_NONE; // ring value of the point's cell: >= 0 pointer (ring index), < 0 resolved (-1 - slot, AB_NONE, AB_DEAD)
            int ri = 0;      // ring index of the point's cell
            if (act)
            {
                const int row = (int) ((meta >> 16) & 127u);
                loc = cidx | (row << 6);
                const long long gc = gc0 + cidx;
                const int cbase = base + H.col_base[cidx];
                const int term = (int) (short) (meta & 0xffffu);
                int v = AB_NONE;
                if (term >= 256)
                    v = (int) ((gc - (term >> 8)) & (RING - 1)) * R + (term & 0xff);
                else if (term >= 0)
                    v = -1 - (cbase + term);
                if ((meta >> 26) & 1u)
                {
                    int lc = lc0 + cidx;
                    lc = lc >= RC ? lc - RC : lc;
                    const int sl = cbase + term;
                    G.g_cell[sl] = lc * R + row;
                    G.birth[sl] = cidx;
                }
                a = v;
                ri = (int) (gc & (RING - 1)) * R + row;
                ring[ri] = (short) v;
            }
            AB_PHW(0)
            ab_barrier(); // B1
            AB_PHW(1)
            // ---------------------------------------------------------------------------------- pointer jumping (one barrier per round)
            for (int r = 0; r < 8; r++)
            {
                int pending = 0;
                if (a >= 0)
                {
                    int bb = ring[a];
#pragma unroll
                    for (int h = 1; h < AB_HOPS; h++) // (several hops per round: a barrier costs more than an LDS round trip)
                        if (bb >= 0)
                            bb = ring[bb];
                    a = bb;
                    ring[ri] = (short) bb;
                    pending = bb >= 0 ? 1 : 0;
                }
                if (__any(pending) && lane == 0)
                    G.jflag[r] = 1;
                ab_barrier();
                if (!G.jflag[r])
                    break;
            }
            AB_PHW(2)
            // ================================================================================== 2: records, alive words, links
            int me = -1;
            int bad = 0;
            if (act)
            {
                if (a <= AB_NONE)
                    bad = 1; // the chain ends in a finished tree (attach refused, cc.cpp:658) or in a cell without a tree
                else
                    me = -1 - a;
            }
            if (!__any(bad))
            {
                // ---- records: one per RUN of consecutive points of a column that lead to the same tree (a tree that shows up in several runs of a
                //      column gets several records: the atomics below merge them). head = first lane of the lane's run, from a prefix maximum over the
                //      lanes that start a run; the run's largest contribution by a segmented DPP maximum; its length from the lane numbers
                {
                    const int key = me >= 0 ? cidx * 256 + me : -1;
                    const int below = dpp_mov_i32<0x138, 0xf>(-2, key); // wave_shr:1 (lane 0 keeps the fill)
                    const int above = dpp_mov_i32<0x130, 0xf>(-2, key); // wave_shl:1 (lane 63 keeps the fill)
                    const bool starts = lane == 0 || below != key;
                    const int head = wave_incl_max_i32(starts ? lane : 0);
                    const unsigned long long mx = ab_run_max_f64_bits(me >= 0 ? pfin : 0ull, lane, head);
                    const bool tail = me >= 0 && (lane == 63 || above != key);
                    if (tail)
                    {
                        atomicMax(&G.g_fin[me], mx);
                        atomicAdd(&G.g_pts[me], (unsigned) (lane - head + 1));
                        atomicMax(&G.g_last[me], cidx);
                    }
                    // the alive word of every record: the columns (lanes) at or behind its column whose smallest azimuth the record's value exceeds
                    unsigned long long tm = __ballot(tail);
                    while (tm)
                    {
                        const int t = __builtin_ctzll(tm);
                        tm &= tm - 1ull;
                        const double f = __longlong_as_double(lane_i64((long long) mx, t));
                        const int st_ = __builtin_amdgcn_readlane(me, t);
                        const int ct = __builtin_amdgcn_readlane(cidx, t);
                        const unsigned long long mm = __ballot(lane >= ct && f > mz);
                        if (lane == 0 && mm)
                            atomicOr(&G.g_alive[st_], mm);
                    }
                }
                AB_PHW(3)
                // ---- link candidates (accepted candidates after the first, cc.cpp:693-694) that lead to another tree: the (up to four) slot-ring
                //      entries of a point are read at once; nearly all of them name the point's own tree
                {
                    const int nlc = (int) ((meta >> 23) & 7u);
                    const int n = (me >= 0 && nlc != 7) ? nlc : 0;
                    if (__any(n > 0))
                    {
                        const long long gc = gc0 + cidx;
                        int other[LINK_SLOTS];
                        bool any_other = false;
#pragma unroll
                        for (int j = 0; j < LINK_SLOTS; j++)
                        {
                            const int code = (int) ((plk >> (16 * j)) & 0xffff);
                            // (a lane without a j-th candidate reads its own cell)
                            const int rj = j < n ? (int) ((gc - (code >> 8)) & (RING - 1)) * R + (code & 0xff) : ri;
                            const int v = ring[rj];
                            other[j] = -1;
                            if (j < n)
                            {
                                if (v >= 0)
                                    bad = 1; // (cannot happen: every cell of the group is resolved)
                                else if (v > AB_NONE && -1 - v != me)
                                {
                                    other[j] = -1 - v;
                                    any_other = true;
                                }
                            }
                        }
                        // rare (two trees meet): column by column, so that a pair of trees is reported once per column (and wavefront) and with that column
                        unsigned long long todo = __ballot(any_other);
                        while (todo)
                        {
                            const int cc = __builtin_amdgcn_readlane(cidx, __builtin_ctzll(todo));
                            const bool mine = any_other && cidx == cc;
                            todo &= ~__ballot(mine);
                            pair[lane] = 0ull;
                            wave_lds_fence();
                            if (mine)
                            {
#pragma unroll
                                for (int j = 0; j < LINK_SLOTS; j++)
                                    if (other[j] >= 0)
                                    {
                                        const unsigned long long bit = 1ull << other[j];
                                        if (!(atomicOr(&pair[me], bit) & bit))
                                        {
                                            const int e = atomicAdd(&G.n_ev, 1);
                                            if (e < AB_EVENTS)
                                                G.ev[e] = ((unsigned) cidx << 16) | ((unsigned) me << 8) | (unsigned) other[j];
                                        }
                                    }
                            }
                            wave_lds_fence();
                        }
                    }
                }
                AB_PHW(4)
            }
            if (__any(bad) && lane == 0)
                G.wbail = AB_BAIL_DEAD;
            if (lane == 0 && lds_ld(&G.n_ev) > AB_EVENTS)
                atomicMax(&G.wbail, (int) AB_BAIL_LINKS); // (never over an attach to a finished tree: DEAD > LINKS)
            ab_barrier(); // B2
            AB_PHW(5)
#if defined(CC_AB_STATS) && defined(CC_AB_STATS_W)
            ab_t[7]++;
#endif
            // ---- the common decision (same LDS words as the timeline's)
            if (i > 0 && T.tbail != 0)
            {
                why_stop = 1;
                break;
            }
            // (the pass of group i - 1 committed: its roots are owed; they are written at the top of the next iteration or behind the loop)
            loc2 = loc1, a2 = a1;
            loc1 = loc, a1 = a;
            g2_gc0 = g1_gc0, g2_lc0 = g1_lc0;
            g1_gc0 = gc0, g1_lc0 = lc0;
            if (ncols == 0 || G.wbail)
            {
                why_stop = 2;
                last_tb = b ^ 1; // (the pass of group i - 1)
                break;
            }
        }
        // roots of the last committed group (after the rotation above it is the older set: the newer one is the group that was not taken / does not exist)
        if (why_stop == 2)
            write_roots(last_tb, loc2, a2, g2_gc0, g2_lc0);
    }

    // ---- persist the tree state back to the global planes (the serial kernels and the next batch load it from there) ------------------
#ifdef CC_AB_STATS
    ab_t[14] = __builtin_amdgcn_s_memtime() - ab_entry; // entry -> end of the loop, as this wavefront saw it
    ab_t[15] = wall_clock64() - ab_entry_rt;
#endif
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
            const long long gc_final = s_gc_final;
            const int reason = s_reason;
            const bool bailed = reason != 0;
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
            st->batch[slot].acp_next = gc_final;
            // a limited launch of the serial kernel takes the group that could not be taken here (and no more), then this kernel is tried again
                        st->serial_until = bailed ? gc_final + s_serial_cols : 0;
            st->batch_columns += s_cols;
            st->batch_bails += bailed ? 1ull : 0ull;
            if (bailed)
            {
                st->batch_bail_reason[reason & 7] += 1ull;
                if (bail_count)
                    atomicAdd(bail_count, 1); // (the engine launches more (batch-parallel, serial) rounds for the next batches: assoc_rounds 0)
            }
            st->n_events = n_events < g.event_capacity ? n_events : g.event_capacity;
            if (g.record_events && n_events > g.event_capacity)
                raise_error(st, CC_ERR_CAPACITY, n_events, 0);
#if defined(CC_AB_STATS) && !defined(CC_AB_STATS_W)
            ab_t[9] = __builtin_amdgcn_s_memtime() - ab_t0;
            for (int i = 0; i < 16; i++)
                st->dbg[i] += ab_t[i];
#endif
        }
    }
#if defined(CC_AB_STATS) && defined(CC_AB_STATS_W)
    if (wave == CC_AB_STATS_WAVE && lane == 0)
    {
        ab_t[9] = __builtin_amdgcn_s_memtime() - ab_t0;
        for (int i = 0; i < 16; i++)
            st->dbg[i] += ab_t[i];
    }
#endif
}

template<int RPL>
__global__ __launch_bounds__(AB_THREADS, CC_AB_MIN_WAVES_PER_SIMD) void k_assocb(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot,
                                                       int* __restrict__ bail_count)
{
    __shared__ AbShared<RPL> S;
    assocb_body<RPL>(g, cfg, P, states, first_stream + (int) blockIdx.x, slot, bail_count, S, AbPreloaded{0, 0, 0, 0});
}

