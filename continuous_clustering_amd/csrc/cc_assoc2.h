// Two-wave association kernel (included by cc_kernels.h inside namespace cck).
//
// k_assoc2 does the work of k_assoc_lds (association bookkeeping, union-find, finished-cluster check, publishing; cc.cpp:643-696,
// 773-1092) with the column recurrence split over two wavefronts of one block, because a lone wavefront issues at most one
// instruction every 4 cycles and every dependent LDS access costs it ~55 cycles (tools/ubench/lone_wave.hip):
//
//   wave A ("front")  per column: which tree does every point of the column join? Needs only k_scan's parent codes and the ids
//                     of the previous columns (the s_win ring), so it never touches the tree state and runs ahead of wave B.
//   wave B ("back")   per column: checks what A assumed (tree still unfinished, one-rotation limit, first unpublished column),
//                     creates the new trees, applies the point / link updates to the tree state, runs the finished-cluster check
//                     and the publish bookkeeping — everything whose order defines the reference's results.
//
// A runs ahead speculatively: a tree finished by B at column c may still be joined by A in columns > c. B sees that when it
// reaches such a column (the tree is dead), parks A, replays the column with the exact serial routine and restarts A behind it.
// Tree ids are stable (no compaction, so nothing A wrote is renumbered); a freed id is quarantined until no ring entry can
// still name it.
#pragma once

constexpr int WIN2_COLS = 64;      // ring of per-cell tree ids: WIN_COLS of look-back + the lead of wave A
constexpr int A2_LEAD = 24;        // columns wave A may run ahead of wave B (WIN_COLS + A2_LEAD + 1 <= WIN2_COLS)
constexpr int A2_INFO = 32;        // per-column hand-off records (power of two > A2_LEAD)
constexpr int A2_FRESH = 0x4000;   // s_win entry flag: the point's tree starts in this very column
constexpr int A2_IDMASK = 0x3fff;
constexpr int A2_SPIN_LIMIT = 1 << 20; // ~30 ms of polling: a broken hand-shake raises an error instead of hanging
enum
{
    A2_RUN = 0,
    A2_PARK = 1,
    A2_EXIT = 2
};

struct LdsTrees2
{
    int cell[TREE_SLOTS];                 // root cell of tree id i
    long long gcol[TREE_SLOTS];           // its global column
    unsigned long long fin[TREE_SLOTS];   // bits of finished_at_continuous_azimuth_angle (non-negative double)
    unsigned last[TREE_SLOTS];            // low 32 bits of the last global column that attached a point
    unsigned pts[TREE_SLOTS];
    int uf[TREE_SLOTS];                   // union-find parent (tree id)
    unsigned long long c_fin[TREE_SLOTS]; // at a representative: lower bound of the cluster's max finished_at
    short alist[TREE_SLOTS];              // ids of the unfinished trees in creation order (the reference's sc_unfinished_point_trees_)
    unsigned char alive[TREE_SLOTS];      // 1: unfinished tree
    // finish check scratch
    unsigned long long a_fin[TREE_SLOTS];
    long long a_min[TREE_SLOTS];
    long long a_max[TREE_SLOTS];
    unsigned a_pts[TREE_SLOTS];
    unsigned a_first[TREE_SLOTS];
    unsigned a_cid[TREE_SLOTS];
    int comp[TREE_SLOTS];
    unsigned char a_flag[TREE_SLOTS];
    // FIFO of free ids (head: consumer = wave A, or wave B while A is parked; tail: wave B)
    short ring_id[TREE_SLOTS];
    long long ring_rel[TREE_SLOTS]; // first column at which the id may be handed out again
    // per-column hand-off A -> B
    int info_head[A2_INFO]; // ring head before the column's allocations
    int info_bad[A2_INFO];  // 1: A could not resolve the column (a candidate without a live id), 2: out of ids
    // control
    long long a_done;       // columns < a_done are resolved
    long long b_done;       // columns < b_done are fully processed
    long long restart_col;
    int cmd;                // A2_RUN / A2_PARK / A2_EXIT (written by B)
    int a_parked;
    int head;               // valid while A is parked
    int tail;
    int bcast_i[4];
    double bcast_d[2];
    long long bcast_l[2];
};

__device__ __forceinline__ bool cluster_may_finish2(LdsTrees2& T, int n_unf, double min_az, double& lower_bound)
{
    bool may = false;
    double lb = 1.7976931348623157e308;
    for (int k = lane_id(); k < n_unf; k += 64)
    {
        const int i = T.alist[k];
        if (lds_ld(&T.uf[i]) == i)
        {
            const double f = __longlong_as_double((long long) lds_ld(&T.c_fin[i]));
            may |= !(f > min_az);
            lb = f < lb ? f : lb;
        }
    }
    lower_bound = uniform_f64(wave_min_f64(lb));
    return __any(may);
}

// exact single-lane replay of one column (rare): reference semantics with immediate attach / link; ids come from the free ring
template<int RPL>
__device__ void assoc_column_live2(const AssocCtx& c, const cc_config& cfg, const Geometry& g, LdsTrees2& T, short* s_win, const int lc,
                                   const long long gc, const int first_local, int& n_unf, double& L, long long& M, int& head, int& err)
{
    const SP& p = c.p;
    const int R = c.R, RC = c.RC;
    short* wcol = s_win + (int) (gc & (WIN2_COLS - 1)) * R;
    for (int row = 0; row < R; row++)
        wcol[row] = -1;
    for (int row = 0; row < R; row++)
    {
        const int pi = lc * R + row;
        if (p.ignored[pi])
        {
            p.root[pi] = -1;
            continue;
        }
        const float mad = ccm::asinf_exact(cfg.max_distance / p.dist[pi]);
        const double pcaz = p.caz[pi];
        const float pincl = p.incl[pi], px = p.x[pi], py = p.y[pi], pz = p.z[pi];
        int needed = f2i_x86(__builtin_ceilf(mad / c.az_width));
        needed = needed < c.max_steps_in_row ? needed : c.max_steps_in_row;
        int oc = lc;
        long long ogc = gc;
        int pslot = -1; // tree id of the point (-1: none yet)
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
                    if (ccm::absf(p.incl[oi] - pincl) > mad)
                        break;
                    if (!p.ignored[oi])
                    {
                        int oslot = s_win[(int) (ogc & (WIN2_COLS - 1)) * R + orow];
                        oslot = oslot < 0 ? oslot : (oslot & A2_IDMASK);
                        if (oslot >= 0 && !T.alive[oslot])
                            oslot = -2; // finished tree
                        // cc.cpp:733: same root -> skip, unless the point's root sits in local column 0 (reference quirk; a
                        // same-tree candidate then only produces a self link, which is a no-op here)
                        const bool same = pslot >= 0 && oslot == pslot;
                        if (!same)
                        {
                            const float dx = px - p.x[oi], dy = py - p.y[oi], dz = pz - p.z[oi];
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
                                    lds_union(T.uf, T.c_fin, pslot, oslot);
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
        if (pslot == -1)
        {
            if (T.tail - head < 1 || T.ring_rel[head & (TREE_SLOTS - 1)] > gc)
            {
                err = CC_ERR_CAPACITY; // out of tree ids mid-column: this kernel cannot roll the column back
                return;
            }
            pslot = T.ring_id[head & (TREE_SLOTS - 1)];
            head++;
            const double fin = pcaz + (double) mad;
            T.cell[pslot] = pi;
            T.gcol[pslot] = gc;
            T.fin[pslot] = (unsigned long long) __double_as_longlong(fin);
            T.last[pslot] = (unsigned) gc;
            T.pts[pslot] = 1;
            T.uf[pslot] = pslot;
            T.c_fin[pslot] = T.fin[pslot];
            T.alist[n_unf] = (short) pslot;
            T.alive[pslot] = 1;
            if (n_unf == 0)
                M = gc;
            n_unf++;
            L = fin < L ? fin : L;
        }
        wcol[row] = (short) pslot;
        p.root[pi] = T.cell[pslot];
    }
}

// one-section-at-a-time cycle probes of wave B (each s_memtime pair costs ~100 cycles, so only the section selected at build time
// with -DCC_A2_SECTION=k is timed; tools/prof_assoc2.py loops over k)
#ifdef CC_A2_SECTION
#define A2_T(k)                  \
    if ((k) == CC_A2_SECTION)    \
        a2_t0 = __builtin_amdgcn_s_memtime();
#define A2_E(k)                  \
    if ((k) == CC_A2_SECTION)    \
        a2_acc += __builtin_amdgcn_s_memtime() - a2_t0;
#else
#define A2_T(k)
#define A2_E(k)
#endif

template<int RPL>
__global__ __launch_bounds__(128) void k_assoc2(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot)
{
    const int s = first_stream + blockIdx.x;
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    StreamState* st = &states[s];
    if (st->error != 0 || st->batch[slot].seg_begin < 0 || st->assoc_mode != 0 || st->batch[slot].acp_next >= st->batch[slot].seg_end)
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

    __shared__ LdsTrees2 T;
    __shared__ short s_win[WIN2_COLS * WAVE * RPL];
    __shared__ int s_parent[WAVE * RPL];
    __shared__ int s_newslot[WAVE * RPL];

    const long long col_begin = st->batch[slot].acp_next, col_end = st->batch[slot].seg_end, first_column = st->first_column;
    const int n_unf0 = st->n_unfinished;
    const int tree_limit = g.lds_tree_limit;
    if (n_unf0 > tree_limit)
    {
        if (threadIdx.x == 0)
        {
            if (st->batch[slot].pub_begin < 0)
                st->batch[slot].pub_begin = st->first_unpublished;
            st->batch[slot].pub_end = st->first_unpublished;
            st->assoc_mode = 1; // the global-memory kernel continues this stream
        }
        return;
    }

    // ---- load the persistent tree state (global planes indexed by root cell): id = list position ------------------------------
    for (int i = threadIdx.x; i < TREE_SLOTS; i += 128)
    {
        T.alive[i] = 0;
        if (i < n_unf0)
        {
            const int cell = p.ulist[i];
            const long long tg = p.colg[cell / R];
            T.cell[i] = cell;
            T.gcol[i] = tg;
            T.fin[i] = (unsigned long long) __double_as_longlong(p.t_fin[cell]);
            T.last[i] = (unsigned) tg + p.t_width[cell] - 1u;
            T.pts[i] = p.t_pts[cell];
            T.uf[i] = p.t_pos[p.t_uf[cell]];
            T.c_fin[i] = T.fin[i];
            T.alist[i] = (short) i;
            T.alive[i] = 1;
        }
        else
        {
            T.ring_id[i - n_unf0] = (short) i;
            T.ring_rel[i - n_unf0] = -0x7fffffffffffffffll;
        }
    }
    if (threadIdx.x == 0)
    {
        T.a_done = col_begin;
        T.b_done = col_begin;
        T.restart_col = col_begin;
        T.cmd = A2_RUN;
        T.a_parked = 0;
        T.head = 0;
        T.tail = TREE_SLOTS - n_unf0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_unf0; i += 128)
        atomicMax(&T.c_fin[lds_find(T.uf, i)], T.fin[i]);
    {
        // ring of tree ids for the WIN2_COLS columns before col_begin (only the last WIN_COLS can be looked at): two dependent
        // gathers per cell (root plane, then the tree planes at the root), 8 cells at a time
        constexpr int B = 8;
        for (int i0 = threadIdx.x; i0 < WIN2_COLS * R; i0 += 128 * B)
        {
            int rr[B];
#pragma unroll
            for (int u = 0; u < B; u++)
            {
                const int i = i0 + u * 128;
                rr[u] = -1;
                if (i < WIN2_COLS * R)
                {
                    const int wc = i / R, row = i - wc * R;
                    // the global column in [col_begin - WIN2_COLS, col_begin) that maps to ring column wc
                    const long long gcx = col_begin - 1 - (((col_begin - 1) % WIN2_COLS - wc + WIN2_COLS) % WIN2_COLS);
                    if (gcx >= first_column && gcx >= 0 && first_column >= 0 && col_begin - gcx <= WIN_COLS)
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
                const int i = i0 + u * 128;
                if (i < WIN2_COLS * R)
                    s_win[i] = (short) (rr[u] < 0 ? -1 : (fin_[u] ? -2 : pos_[u]));
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(3); // latency-critical serial chains

    if (wave == 0)
    {
        // =========================================================================================== wave A: resolve
        int head = 0;
        long long gcA = col_begin;
        int lc = (int) (col_begin % RC);
        long long b_seen = col_begin;
        int nx_parent[RPL];
        auto load_parent = [&](long long gcx, int lcx)
        {
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                nx_parent[k] = -2;
                if (row < R && gcx < col_end)
                    nx_parent[k] = p.sc_parent[lcx * R + row];
            }
        };
        load_parent(gcA, lc);
        bool wait_park = false; // a column could not be resolved: wave B will park us when it gets there
#ifdef CC_A2_STATS
        unsigned long long a_lead_waits = 0, a_t0 = __builtin_amdgcn_s_memtime(), a_busy = 0;
#endif
        while (true)
        {
            const int cmd = uniform_i32(lds_ld(&T.cmd)); // every flag read is made wave-uniform: a divergent loop condition would
                                                         // drag all of the wave's scalar bookkeeping into VGPRs
            if (cmd == A2_EXIT)
                break;
            if (cmd == A2_PARK)
            {
                if (lane == 0)
                    lds_st(&T.a_parked, 1);
                while (uniform_i32(lds_ld(&T.cmd)) == A2_PARK)
                    __builtin_amdgcn_s_sleep(1);
                if (uniform_i32(lds_ld(&T.cmd)) == A2_EXIT)
                    break;
                wave_lds_fence();
                gcA = uniform_i64(lds_ld(&T.restart_col));
                head = uniform_i32(lds_ld(&T.head));
                lc = (int) (gcA % RC);
                b_seen = gcA;
                wait_park = false;
                load_parent(gcA, lc);
                continue;
            }
            if (wait_park || gcA >= col_end)
            {
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            if (gcA - b_seen >= A2_LEAD)
            {
                b_seen = uniform_i64(lds_ld(&T.b_done));
                if (gcA - b_seen >= A2_LEAD)
                {
#ifdef CC_A2_STATS
                    a_lead_waits++;
#endif
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
            }
            int parent[RPL];
#pragma unroll
            for (int k = 0; k < RPL; k++)
                parent[k] = nx_parent[k];
            {
                const int lc1 = lc + 1 == RC ? 0 : lc + 1;
                load_parent(gcA + 1, lc1); // prefetch
            }
            const int wcur = (int) (gcA & (WIN2_COLS - 1));
            int cnt_new = 0;
            int newidx[RPL];
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                const bool is_new = parent[k] == -1;
                const unsigned long long mask = __ballot(is_new);
                newidx[k] = cnt_new + __popcll(mask & lanes_below());
                cnt_new += __popcll(mask);
                if (row < R && RPL > 1)
                {
                    const bool same_col = parent[k] >= 0 && (parent[k] >> 8) == 0;
                    s_parent[row] = same_col ? (parent[k] & 0xff) : row;
                    s_newslot[row] = is_new ? newidx[k] : (parent[k] >= 0 ? -1 - parent[k] : 0x7fffffff);
                }
            }
            // ids for the new trees
            int bad = 0;
            if (cnt_new > 0)
            {
                const int tail = uniform_i32(lds_ld(&T.tail));
                if (tail - head < cnt_new || uniform_i64(lds_ld(&T.ring_rel[(head + cnt_new - 1) & (TREE_SLOTS - 1)])) > gcA)
                    bad = 2;
                wave_lds_fence();
            }
            int top_of[RPL];
            if (RPL == 1)
            {
                const bool same_col = parent[0] >= 0 && (parent[0] >> 8) == 0;
                const int prow = parent[0] & 0xff;
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
                wave_lds_fence();
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
                const int mine = parent[0] == -1 ? newidx[0] : (parent[0] >= 0 ? -1 - parent[0] : 0x7fffffff);
                term_info = __shfl(mine, top_of[0]);
            }
            int ent[RPL];
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                ent[k] = -1;
                if (parent[k] >= -1 && row < R)
                {
                    const int tv = RPL == 1 ? term_info : s_newslot[top_of[k]];
                    if (tv >= 0)
                    {
                        if (bad == 0)
                            ent[k] = (int) T.ring_id[(head + tv) & (TREE_SLOTS - 1)] | A2_FRESH;
                    }
                    else
                    {
                        const int code = -1 - tv;
                        const int delta = code >> 8, prow = code & 0xff;
                        const int v = s_win[((wcur - delta) & (WIN2_COLS - 1)) * R + prow];
                        if (v < 0)
                            bad = bad ? bad : 1; // no tree, or a tree finished before this launch: the exact routine decides
                        else
                            ent[k] = v & A2_IDMASK;
                    }
                }
            }
            bad = uniform_i32(__any(bad == 2) ? 2 : (__any(bad == 1) ? 1 : 0));
            short* wcol = s_win + wcur * R;
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (row < R)
                    wcol[row] = (short) ent[k];
            }
            if (lane == 0)
            {
                T.info_head[(int) (gcA & (A2_INFO - 1))] = head;
                T.info_bad[(int) (gcA & (A2_INFO - 1))] = bad;
            }
            wave_lds_fence();
            if (lane == 0)
                lds_st(&T.a_done, gcA + 1);
            if (bad)
                wait_park = true;
            else
                head += cnt_new;
            gcA++;
            lc = lc + 1 == RC ? 0 : lc + 1;
        }
#ifdef CC_A2_STATS
        if (lane == 0)
        {
            st->dbg[9] += a_lead_waits;
            st->dbg[11] += __builtin_amdgcn_s_memtime() - a_t0;
        }
#endif
        return;
    }

    // ================================================================================================= wave B: apply + finish
    long long first_unpub = st->first_unpublished, ring_start = st->ring_start;
    if (lane == 0 && st->batch[slot].pub_begin < 0)
        st->batch[slot].pub_begin = first_unpub; // first association kernel of this pass
    unsigned long long cluster_counter = st->cluster_counter;
    int n_unf = n_unf0;
    long long M = st->min_required;
    double L = st->finish_lower_bound;
    double last_min_az = st->last_round_min_az;
    unsigned long long cells_published = st->cells_published, clusters_finished = st->clusters_finished;
    unsigned long long exceed = st->exceed_one_rotation, serial_cols = st->serial_columns, alias_rounds = st->stamp_alias_rounds;
    int n_events = st->n_events;
    int err = 0;
    long long err_a = 0, err_b = 0;
    bool to_global = false;

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
        // lane 0 only: a divergent (vector) load; a uniform one would become a scalar load that every LDS wait has to sit out
        if (lane == 0 && gcx < col_end)
            nx_minaz = p.colminaz[lcx];
    };
    int lc = (int) (col_begin % RC);
    int nth_phase = (int) (col_begin % nth);
    long long first_local_of = first_unpub;
    int first_local = (int) (first_unpub % RC);
    load_column(col_begin, lc);
    long long a_seen = col_begin;

    // park wave A, run `body` with exclusive access to the id ring, restart A at `restart`
    auto park_a = [&]()
    {
        if (lane == 0)
            lds_st(&T.cmd, (int) A2_PARK);
        int spins = 0;
        while (uniform_i32(lds_ld(&T.a_parked)) == 0)
        {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > A2_SPIN_LIMIT)
            {
                err = CC_ERR_BOOKKEEPING; // hand-shake broken: fail loudly instead of hanging the device
                err_a = -771;
                break;
            }
        }
    };
    auto resume_a = [&](long long restart, int head)
    {
        if (lane == 0)
        {
            T.restart_col = restart;
            T.head = head;
            T.a_done = restart; // what A resolved beyond this column is void
            T.a_parked = 0;
        }
        wave_lds_fence();
        if (lane == 0)
            lds_st(&T.cmd, (int) A2_RUN);
        a_seen = restart;
    };

#ifdef CC_A2_SECTION
    unsigned long long a2_t0 = 0, a2_acc = 0;
#endif
#ifdef CC_A2_STATS
    unsigned long long b_waits = 0, b_t0 = __builtin_amdgcn_s_memtime();
#endif
    long long gc = col_begin;
    for (; gc < col_end && err == 0; gc++, lc = (lc + 1 == RC ? 0 : lc + 1), nth_phase = (nth_phase + 1 == nth ? 0 : nth_phase + 1))
    {
        A2_E(6)
        A2_T(6)
        A2_T(0)
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
        A2_E(0)
        A2_T(1)
        load_column(gc + 1, lc + 1 == RC ? 0 : lc + 1); // prefetch: nothing below depends on it
        for (int spins = 0; a_seen <= gc;)
        {
            a_seen = uniform_i64(lds_ld(&T.a_done));
            if (a_seen <= gc)
            {
#ifdef CC_A2_STATS
                b_waits++;
#endif
                __builtin_amdgcn_s_sleep(1);
                if (++spins > A2_SPIN_LIMIT)
                {
                    err = CC_ERR_BOOKKEEPING;
                    err_a = -772;
                    err_b = gc;
                    break;
                }
            }
        }
        if (err)
            break;
        wave_lds_fence(); // the column's ring entries are read after the flag
        const int wcur = (int) (gc & (WIN2_COLS - 1));
        short* wcol = s_win + wcur * R;
        const int info_bad = uniform_i32(lds_ld(&T.info_bad[(int) (gc & (A2_INFO - 1))]));
        const int info_head = uniform_i32(lds_ld(&T.info_head[(int) (gc & (A2_INFO - 1))]));

        A2_E(1)
        A2_T(2)
        int cnt_new = 0;
        int newrank[RPL];
        bool bad = info_bad != 0;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const unsigned long long mask = __ballot(parent[k] == -1);
            newrank[k] = cnt_new + __popcll(mask & lanes_below());
            cnt_new += __popcll(mask);
            if (nl[k] == 255)
                bad = true; // more links than the scan records: exact routine
        }
        if (n_unf + cnt_new > tree_limit || info_bad == 2)
        {
            to_global = true; // continue this stream with the global-memory kernel, starting at this column
            break;
        }
        emit(CC_EV_GROUND_COLUMN, gc, gc, 0, 0, gc);

        // ---- what wave A assumed (cc.cpp:657-658, 762-763) -------------------------------------------------------------------
        int id[RPL];
        const bool span_check = n_unf > 0 && (uint32_t) (gc - M + 1) > (uint32_t) NC;
        const bool reach_check = gc - (WIN_COLS - 1) < first_unpub;
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            const int row = k * 64 + lane;
            id[k] = -1;
            if (!bad && parent[k] >= -1 && row < R)
            {
                const int e = wcol[row];
                id[k] = e & A2_IDMASK;
                if (!(e & A2_FRESH))
                {
                    if (!T.alive[id[k]])
                        bad = true; // finished tree: attach refused (cc.cpp:658)
                    else if (span_check && (uint32_t) (gc - T.gcol[id[k]] + 1) > (uint32_t) NC)
                        bad = true; // tree would span more than one rotation (cc.cpp:657)
                }
                if (reach_check)
                {
                    // nothing may come from columns the live scan would not have reached (cc.cpp:762-763)
                    int oldest_delta = 0;
                    if (parent[k] >= 0)
                    {
                        oldest_delta = parent[k] >> 8;
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
        // (the terminal of a same-column chain is checked by the chain's top row: its own parent code carries that delta)
        const bool column_live = __any(bad);
        A2_E(2)

        if (!column_live)
        {
            A2_T(3)
            double l_new = L;
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                if (row < R && parent[k] == -1)
                {
                    const int i = id[k];
                    T.cell[i] = lc * R + row;
                    T.gcol[i] = gc;
                    T.fin[i] = (unsigned long long) __double_as_longlong(finc[k]);
                    T.last[i] = (unsigned) gc;
                    T.pts[i] = 1;
                    T.uf[i] = i;
                    T.c_fin[i] = (unsigned long long) __double_as_longlong(finc[k]);
                    T.alist[n_unf + newrank[k]] = (short) i;
                    T.alive[i] = 1;
                    l_new = finc[k] < l_new ? finc[k] : l_new;
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
                const int row = k * 64 + lane;
                if (row < R)
                    p.root[lc * R + row] = id[k] >= 0 ? T.cell[id[k]] : -1; // early: retires long before the next loop-top wait
            }
            A2_E(3)
            A2_T(4)
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                if (parent[k] >= 0)
                {
                    const int i = id[k];
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
                            int v = s_win[((wcur - (code >> 8)) & (WIN2_COLS - 1)) * R + (code & 0xff)];
                            v = v < 0 ? v : (v & A2_IDMASK);
                            if (v >= 0 && v != i && T.alive[v])
                                lds_union(T.uf, T.c_fin, i, v);
                        }
                }
            }
            wave_lds_fence();
            A2_E(4)
        }
        else
        {
            serial_cols++;
            park_a();
            if (err)
                break;
            if (lane == 0)
            {
                int nn = n_unf, e = 0, hd = info_head;
                double LL = L;
                long long MM = M;
                assoc_column_live2<RPL>(c, cfg, g, T, s_win, lc, gc, first_local, nn, LL, MM, hd, e);
                T.bcast_i[0] = nn;
                T.bcast_i[1] = e;
                T.bcast_i[3] = hd;
                T.bcast_d[0] = LL;
                T.bcast_l[0] = MM;
            }
            wave_lds_fence();
            n_unf = uniform_i32(T.bcast_i[0]);
            const int hd = uniform_i32(T.bcast_i[3]);
            if (uniform_i32(T.bcast_i[1]) == CC_ERR_CAPACITY)
            {
                err = CC_ERR_CAPACITY;
                err_a = n_unf;
            }
            L = uniform_f64(T.bcast_d[0]);
            M = uniform_i64(T.bcast_l[0]);
            wave_lds_fence();
            if (!err)
                resume_a(gc + 1, hd);
        }
        if (err)
            break;

        // ------------------------------------------------------------------ finished-cluster check (cc.cpp:837-974)
        if (nth_phase != 0)
        {
            if (lane == 0)
                lds_st(&T.b_done, gc + 1);
            continue;
        }
        A2_T(5)
        long long M_c;
        if (n_unf == 0)
            M_c = gc + 1;
        else if (min_az == last_min_az)
        {
            alias_rounds++;
            M_c = M;
        }
        else if (!((gc + 1 - M) >= NC) && (!(min_az >= L) || !cluster_may_finish2(T, n_unf, min_az, L)))
            M_c = M; // nothing can be finished: first the scalar bound, then (refreshing it) the per-cluster bounds
        else
        {
            A2_T(7)
            for (int k = lane; k < n_unf; k += 64)
            {
                const int i = T.alist[k];
                T.a_fin[i] = 0ull;
                T.a_min[i] = 0x7fffffffffffffffll;
                T.a_max[i] = 0;
                T.a_pts[i] = 0;
                T.a_first[i] = 0xffffffffu;
                T.a_cid[i] = 0;
                T.a_flag[i] = 0;
            }
            wave_lds_fence();
            for (int k = lane; k < n_unf; k += 64)
            {
                const int i = T.alist[k];
                const int j = lds_find(T.uf, i);
                T.comp[i] = j;
                atomicMax(&T.a_fin[j], T.fin[i]);
                atomicMin(&T.a_min[j], T.gcol[i]);
                atomicMax(&T.a_max[j], T.gcol[i] + (long long) (T.last[i] - (unsigned) T.gcol[i] + 1u));
                atomicAdd(&T.a_pts[j], T.pts[i]);
                atomicMin(&T.a_first[j], (unsigned) k);
            }
            wave_lds_fence();
            int exceed_local = 0;
            for (int k = lane; k < n_unf; k += 64)
            {
                const int i = T.alist[k];
                if (T.comp[i] == i)
                {
                    const double fin = __longlong_as_double((long long) T.a_fin[i]);
                    const bool unfinished = fin > min_az;
                    const bool exceeds = (T.a_max[i] - T.a_min[i]) >= NC;
                    if (exceeds)
                        exceed_local++;
                    T.a_flag[i] = (!unfinished || exceeds) ? 1 : 0;
                }
            }
            for (int o = 32; o > 0; o >>= 1)
                exceed_local += __shfl_xor(exceed_local, o);
            exceed += (unsigned long long) uniform_i32(exceed_local);
            wave_lds_fence();
            // cluster ids in list order of each cluster's first tree (cc.cpp:921-1001 walks the list front to back)
            int last_first = -1;
            while (true)
            {
                int best = 0x7fffffff;
                for (int k = lane; k < n_unf; k += 64)
                {
                    const int i = T.alist[k];
                    if (T.comp[i] == i && T.a_flag[i] && T.a_pts[i] > 5u)
                    {
                        const int fi = (int) T.a_first[i];
                        if (fi > last_first && fi < best)
                            best = fi;
                    }
                }
                best = uniform_i32(wave_min_i32(best));
                if (best == 0x7fffffff)
                    break;
                const int j = T.comp[T.alist[best]];
                const unsigned cid = (unsigned) cluster_counter;
                if (lane == 0)
                    T.a_cid[j] = cid;
                emit(CC_EV_CLUSTER, T.a_min[j], T.a_max[j] - 1, cid, T.a_pts[j], gc);
                cluster_counter++;
                clusters_finished++;
                last_first = best;
            }
            wave_lds_fence();
            // persist + retire the finished trees (their ids return to the ring after the look-back window has passed them);
            // the list of unfinished trees is compacted in place, in order
            long long min_all = 0x7fffffffffffffffll, min_surv = 0x7fffffffffffffffll;
            double L_new = 1.7976931348623157e308;
            int removed = 0, out = 0;
            const int tail = uniform_i32(T.tail);
            for (int base = 0; base < n_unf; base += 64)
            {
                const int k = base + lane;
                bool dead = false, surv = false;
                int i = 0;
                if (k < n_unf)
                {
                    i = T.alist[k];
                    const int j = T.comp[i];
                    const long long tg = T.gcol[i];
                    min_all = tg < min_all ? tg : min_all;
                    if (T.a_flag[j])
                    {
                        const int cell = T.cell[i];
                        p.t_finished[cell] = 1;
                        p.t_cid[cell] = T.a_cid[j];
                        dead = true;
                    }
                    else
                    {
                        surv = true;
                        min_surv = tg < min_surv ? tg : min_surv;
                        T.c_fin[i] = T.a_fin[j]; // exact cluster maximum (only read at representatives)
                        if (j == i)
                        {
                            const double f = __longlong_as_double((long long) T.a_fin[i]);
                            L_new = f < L_new ? f : L_new;
                        }
                    }
                }
                const unsigned long long dmask = __ballot(dead), smask = __ballot(surv);
                wave_lds_fence(); // every read of this block of the list precedes its in-place rewrite
                if (dead)
                {
                    const int pos = tail + removed + __popcll(dmask & lanes_below());
                    T.ring_id[pos & (TREE_SLOTS - 1)] = (short) i;
                    T.ring_rel[pos & (TREE_SLOTS - 1)] = gc + WIN_COLS;
                    T.alive[i] = 0;
                }
                if (surv)
                    T.alist[out + __popcll(smask & lanes_below())] = (short) i;
                removed += __popcll(dmask);
                out += __popcll(smask);
            }
            wave_lds_fence();
            if (removed > 0 && lane == 0)
                lds_st(&T.tail, tail + removed);
            min_all = uniform_i64(wave_min_i64(min_all));
            min_surv = uniform_i64(wave_min_i64(min_surv));
            L = uniform_f64(wave_min_f64(L_new));
            M_c = min_all;
            M = min_surv;
            n_unf -= removed;
            wave_lds_fence();
            A2_E(7)
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
        if (lane == 0)
            lds_st(&T.b_done, gc + 1);
        A2_E(5)
    }
#ifdef CC_A2_SECTION
    if (lane == 0)
        st->dbg[CC_A2_SECTION] += a2_acc;
#endif
#ifdef CC_A2_STATS
    if (lane == 0)
    {
        st->dbg[8] += b_waits;
        st->dbg[10] += __builtin_amdgcn_s_memtime() - b_t0;
        st->dbg[12] += (unsigned long long) (gc - col_begin);
    }
#endif
    if (lane == 0)
        lds_st(&T.cmd, (int) A2_EXIT);

    // ---- persist the tree state back to the global planes: list order = creation order -----------------------------------------
    {
        wave_lds_fence();
        for (int r = lane; r < n_unf; r += 64)
        {
            const int i = T.alist[r];
            T.a_first[i] = (unsigned) r; // list position of every unfinished tree (t_uf names the representative's root cell)
        }
        wave_lds_fence();
        for (int r = lane; r < n_unf; r += 64)
        {
            const int i = T.alist[r];
            const int cell = T.cell[i];
            p.ulist[r] = cell;
            p.t_pos[cell] = r;
            p.t_fin[cell] = __longlong_as_double((long long) T.fin[i]);
            p.t_width[cell] = T.last[i] - (unsigned) T.gcol[i] + 1u;
            p.t_pts[cell] = T.pts[i];
            p.t_uf[cell] = T.cell[T.uf[i]];
            p.t_cid[cell] = 0;
            p.t_finished[cell] = 0;
        }
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
