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
// columns wave A may run ahead of wave B (WIN_COLS + lead + 1 <= WIN2_COLS); two rows per lane: half the lead, half the staging
// (the block has to share the CU's LDS with the 96 KB of k_insert2<2>)
constexpr int a2_lead(int rpl)
{
    return rpl == 1 ? 24 : 12;
}
constexpr int A2_INFO = 32;        // per-column hand-off records (power of two > lead)
// columns of staged per-point data wave A keeps ahead for wave B (power of two >= lead + group size)
constexpr int a2_stage(int rpl)
{
    return rpl == 1 ? 32 : 16;
}
constexpr int A2_FRESH = 0x4000;   // s_win entry flag: the point's tree starts in this very column
constexpr int A2_IDMASK = 0x3fff;
constexpr int A2_SPIN_LIMIT = 1 << 23; // ~0.25 s of polling: a broken hand-shake raises an error instead of hanging
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
    long long last[TREE_SLOTS];           // last global column that attached a point (width = last - gcol + 1)
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
    unsigned long long bcast_u64;
};

// true iff some cluster's (lower-bounded) max finished_at has been passed by the column's minimum azimuth: only then can the
// finished-cluster check of cc.cpp:884-885 let a cluster through. The minimum over the clusters goes through one LDS word
// (non-negative doubles order like their bit patterns): two round trips instead of a 12-step cross-lane reduction.
__device__ __forceinline__ bool cluster_may_finish2(LdsTrees2& T, int n_unf, double min_az, double& lower_bound)
{
    double lb;
    if (n_unf <= 64)
    {
        // the usual case: one tree per lane, minimum by DPP (two LDS round trips, no atomics)
        const int k = lane_id();
        const int i = T.alist[k < n_unf ? k : 0];
        const int rep = lds_ld(&T.uf[i]);
        const unsigned long long f = lds_ld(&T.c_fin[i]);
        lb = uniform_f64(wave_min_f64((k < n_unf && rep == i) ? __longlong_as_double((long long) f) : 1.7976931348623157e308));
    }
    else
    {
        if (lane_id() == 0)
            T.bcast_u64 = 0x7fefffffffffffffull; // DBL_MAX
        wave_lds_fence();
        for (int k = lane_id(); k < n_unf; k += 64)
        {
            const int i = T.alist[k];
            if (lds_ld(&T.uf[i]) == i)
                atomicMin(&T.bcast_u64, lds_ld(&T.c_fin[i]));
        }
        wave_lds_fence();
        lb = uniform_f64(__longlong_as_double((long long) lds_ld(&T.bcast_u64)));
    }
    lower_bound = lb; // min over the clusters of (a lower bound of) their max finished_at
    return !(lb > min_az);
}

// exact single-lane replay of one column (rare): reference semantics with immediate attach / link; ids come from the free ring
template<int RPL>
__device__ void assoc_column_live2(const AssocCtx& c, const cc_config& cfg, const Geometry& g, LdsTrees2& T, short* s_win, const int lc,
                                   const long long gc, const int first_local, int& n_unf, double& L, long long& M, int& head, int& err, StreamState* st)
{
    const SP& p = c.p;
    const int R = c.R, RC = c.RC;
    short* wcol = s_win + (int) (gc & (WIN2_COLS - 1)) * R;
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
                    visits++; // cc.cpp:725
                    const float4 orec = p.sc_rec[oi];
                    if (ccm::absf(orec.w - pincl) > mad)
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
                                            T.last[oslot] = gc;
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
            T.last[pslot] = gc;
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
        p.sc_parent[pi] = (int16_t) parcode; // the live scan's parent replaces the static one
        if (g.mirror_fields)
            p.sc_visits[pi] = sat_u16(visits);
    }
}

// row_shr:N within a row of 16 lanes (lanes without a source keep `fill`): prefix scans over the first lanes without LDS round trips
template<int N>
__device__ __forceinline__ int dpp_shr_i32(int v, int fill)
{
    return __builtin_amdgcn_update_dpp(fill, v, 0x110 + N, 0xf, 0xf, false);
}
template<int N>
__device__ __forceinline__ long long dpp_shr_i64(long long v, long long fill)
{
    const unsigned lo = (unsigned) dpp_shr_i32<N>((int) (unsigned) (unsigned long long) v, (int) (unsigned) (unsigned long long) fill);
    const unsigned hi = (unsigned) dpp_shr_i32<N>((int) (unsigned) ((unsigned long long) v >> 32), (int) (unsigned) ((unsigned long long) fill >> 32));
    return (long long) (((unsigned long long) hi << 32) | lo);
}
template<int N>
__device__ __forceinline__ double dpp_shr_f64(double v, double fill)
{
    return __longlong_as_double(dpp_shr_i64<N>(__double_as_longlong(v), __double_as_longlong(fill)));
}

template<int RPL>
__global__ __launch_bounds__(128) void k_assoc2(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot)
{
    constexpr int G = RPL == 1 ? 8 : 4; // columns wave B handles per pass
    constexpr int A2_LEAD = a2_lead(RPL), A2_STAGE = a2_stage(RPL);
    static_assert(WIN_COLS + A2_LEAD + 1 <= WIN2_COLS && A2_LEAD + G <= A2_STAGE && A2_LEAD < A2_INFO, "ring sizes");
    const int s = first_stream + blockIdx.x;
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
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

    __shared__ LdsTrees2 T;
    __shared__ short s_win[WIN2_COLS * WAVE * RPL];
    // per-point inputs of the columns between the two waves (parent code, finished_at), staged by wave A, which has the time: wave B
    // then issues no per-point global load at all, and its per-column loops are real loops over LDS (small code: the instruction
    // cache is shared and a fully unrolled group body does not fit)
    __shared__ double st_fin[A2_STAGE * WAVE * RPL];
    __shared__ short st_parent[A2_STAGE * WAVE * RPL];

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
            T.last[i] = tg + (long long) p.t_width[cell] - 1;
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
        // Per column: one look-up in the id ring per point (k_scan already followed the same-column parent chains), ids from the
        // free ring for the new roots, the column's ids into the ring. Every flag read is made wave-uniform (readfirstlane): a
        // divergent loop condition would drag all of the wave's scalar bookkeeping into VGPRs.
        int head = 0;
        long long gcA = col_begin;
        int lc = (int) (col_begin % RC);
        long long b_seen = col_begin;
        int nx_term[RPL], nx_info = 0, nx_nl[RPL], nx_par[RPL];
        unsigned long long nx_link[RPL];
        double nx_fin[RPL];
        auto load_a = [&](long long gcx, int lcx)
        {
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                nx_term[k] = -1;
                nx_nl[k] = 0;
                nx_link[k] = 0;
                nx_par[k] = -2;
                nx_fin[k] = 0.;
                if (row < R && gcx < col_end)
                {
                    nx_par[k] = p.sc_parent[lcx * R + row];
                    nx_fin[k] = p.sc_fin[lcx * R + row];
                    nx_term[k] = p.sc_term[lcx * R + row];
                    nx_nl[k] = p.sc_nlinks[lcx * R + row];
                    nx_link[k] = p.sc_links[lcx * R + row]; // (stale where the point has no links: never looked at)
                }
            }
            if (lane == 0 && gcx < col_end)
                nx_info = p.col_info[lcx];
        };
        load_a(gcA, lc);
        bool wait_park = false; // a column could not be resolved: wave B will park us when it gets there
        long long fake_begin = 0, fake_end = 0, fake_group = 0; // columns behind such a column, in wave B's group (see below)
        int poll = 0;
        while (true)
        {
            const bool idle = wait_park || gcA >= col_end || gcA - b_seen >= A2_LEAD;
            if (idle || (++poll & 3) == 0)
            {
                const int cmd = uniform_i32(lds_ld(&T.cmd));
                if (cmd == A2_EXIT)
                    break;
                if (cmd == A2_PARK)
                {
                    // (this wave's tree-root stores of the columns it resolved must have landed before wave B replays one of them)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
                    fake_end = 0;
                    load_a(gcA, lc);
                    continue;
                }
                if (wait_park || gcA >= col_end)
                {
                    if (wait_park && fake_begin < fake_end && uniform_i64(lds_ld(&T.b_done)) >= fake_group)
                    {
                        // wave B works on the group of the column that stopped this wave: nothing older than the group is looked at any more,
                        // the ring entries of the group's remaining columns can be written
                        for (long long x = fake_begin; x < fake_end; x++)
                        {
                            for (int k = 0; k < RPL; k++)
                                if (k * 64 + lane < R)
                                    s_win[(int) (x & (WIN2_COLS - 1)) * R + k * 64 + lane] = -1; // (wave B looks at the ids of the whole group)
                            if (lane == 0)
                            {
                                T.info_head[(int) (x & (A2_INFO - 1))] = head;
                                T.info_bad[(int) (x & (A2_INFO - 1))] = 1;
                            }
                        }
                        wave_lds_fence();
                        if (lane == 0)
                            lds_st(&T.a_done, fake_end);
                        fake_end = 0;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                if (gcA - b_seen >= A2_LEAD)
                {
                    b_seen = uniform_i64(lds_ld(&T.b_done));
                    if (gcA - b_seen >= A2_LEAD)
                    {
                        __builtin_amdgcn_s_sleep(16); // wave B needs thousands of cycles per group: poll rarely
                        continue;
                    }
                }
            }
            int term[RPL], nlk[RPL], parc[RPL];
            unsigned long long lk[RPL];
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                term[k] = nx_term[k];
                nlk[k] = nx_nl[k];
                lk[k] = nx_link[k];
                parc[k] = nx_par[k];
                const int row = k * 64 + lane;
                if (row < R) // stage what wave B needs of this column
                {
                    const int o = (int) (gcA & (A2_STAGE - 1)) * R + row;
                    st_parent[o] = (short) nx_par[k];
                    st_fin[o] = nx_fin[k];
                }
            }
            const int cnt_new = uniform_i32(nx_info) & 0xff;
            const bool col_links = (uniform_i32(nx_info) >> 8) & 2;
            {
                const int lc1 = lc + 1 == RC ? 0 : lc + 1;
                load_a(gcA + 1, lc1); // prefetch
            }
            const int wcur = (int) (gcA & (WIN2_COLS - 1));
            int bad = 0;
            if (cnt_new > 0)
            {
                const int tail = uniform_i32(lds_ld(&T.tail));
                if (tail - head < cnt_new || uniform_i64(lds_ld(&T.ring_rel[(head + cnt_new - 1) & (TREE_SLOTS - 1)])) > gcA)
                    bad = 2;
                wave_lds_fence();
            }
            int ent[RPL];
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                ent[k] = -1;
                const int tm = term[k];
                if (tm >= 256)
                {
                    const int v = s_win[((wcur - (tm >> 8)) & (WIN2_COLS - 1)) * R + (tm & 0xff)];
                    if (v < 0)
                        bad = bad ? bad : 1; // no tree, or a tree finished before this launch: the exact routine decides
                    else
                        ent[k] = v & A2_IDMASK;
                }
                else if (tm >= 0 && bad == 0)
                    ent[k] = (int) T.ring_id[(head + tm) & (TREE_SLOTS - 1)] | A2_FRESH;
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
            // Tree root of every cell (Point::tree_root_, cc.cpp:661,814): the lane of a new root files its cell under the tree id, then
            // every lane reads the root cell of its tree and writes the root plane — this wave has the time, wave B issues no global
            // store per column. A column wave B replays exactly (stale speculation) is rewritten by the replay; what this wave
            // resolves again after a restart is written again.
            if (bad == 0)
            {
#pragma unroll
                for (int k = 0; k < RPL; k++)
                    if (parc[k] == -1 && ent[k] >= 0)
                        T.cell[ent[k] & A2_IDMASK] = lc * R + k * 64 + lane;
                wave_lds_fence();
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    const int cell = T.cell[ent[k] >= 0 ? (ent[k] & A2_IDMASK) : 0];
                    if (row < R)
                        p.root[lc * R + row] = ent[k] >= 0 ? cell : -1;
                }
            }
            // Links (further accepted candidates) only matter where they lead to another tree, which is rare (two trees of one
            // object meeting): this wave, which has the time, looks the targets up and tells wave B whether the column has any.
            int foreign = 0;
            if (col_links && bad == 0)
            {
                wave_lds_fence(); // same-column targets: read what was just written
                bool f = false;
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int mine = ent[k] & A2_IDMASK;
                    int v[LINK_SLOTS];
#pragma unroll
                    for (int j = 0; j < LINK_SLOTS; j++)
                    {
                        v[j] = -1;
                        if (term[k] >= 0 && j < nlk[k])
                        {
                            const int code = (int) ((lk[k] >> (16 * j)) & 0xffff);
                            v[j] = s_win[((wcur - (code >> 8)) & (WIN2_COLS - 1)) * R + (code & 0xff)];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < LINK_SLOTS; j++)
                        f |= v[j] >= 0 && (v[j] & A2_IDMASK) != mine;
                }
                foreign = __any(f) ? 16 : 0;
            }
            if (lane == 0)
            {
                T.info_head[(int) (gcA & (A2_INFO - 1))] = head;
                T.info_bad[(int) (gcA & (A2_INFO - 1))] = bad | foreign;
            }
            wave_lds_fence();
            if (lane == 0)
                lds_st(&T.a_done, gcA + 1);
            if (bad)
            {
                // This wave stops until wave B has replayed the column. Wave B waits for WHOLE groups of columns (wait_a(gc + gcount)): the
                // rest of the column's group is handed over as "not resolved" as well, once wave B has reached the group (see the idle
                // branch above) — waiting for those columns would never end (round 3: streams that attach to trees finished before the
                // launch, or run out of tree ids, in the middle of a group; the spin limit reported error -772).
                wait_park = true;
                const long long group_begin = col_begin + (gcA - col_begin) / G * G;
                const long long group_end = group_begin + G < col_end ? group_begin + G : col_end;
                fake_begin = gcA + 1;
                fake_end = group_end;
                fake_group = group_begin;
            }
            else
                head += cnt_new;
            gcA++;
            lc = lc + 1 == RC ? 0 : lc + 1;
        }
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

    // next group's column summaries (lane u holds column u's), prefetched one group ahead
    double q_minaz = 0., q_newfin = 0.;
    int q_info = 0;
    auto load_group = [&](long long g0, int lcg) // lcg = g0 % RC
    {
        if (lane < G && g0 + lane < col_end)
        {
            int lcl = lcg + lane;
            lcl = lcl >= RC ? lcl - RC : lcl;
            q_minaz = p.colminaz[lcl];
            q_newfin = p.col_newfin[lcl];
            q_info = p.col_info[lcl];
        }
    };
    long long a_seen = col_begin;
#ifdef CC_A2_STATS
    unsigned long long st_wait_g = 0;
#endif

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
    auto wait_a = [&](long long upto) // columns < upto resolved by wave A
    {
        for (int spins = 0; a_seen < upto;)
        {
            a_seen = uniform_i64(lds_ld(&T.a_done));
            if (a_seen < upto)
            {
#ifdef CC_A2_STATS
                st_wait_g++;
#endif
                __builtin_amdgcn_s_sleep(1);
                if (++spins > A2_SPIN_LIMIT)
                {
                    err = CC_ERR_BOOKKEEPING;
                    err_a = -772;
                    err_b = upto;
                    break;
                }
            }
        }
        wave_lds_fence(); // ring entries are read after the flag
    };

#ifdef CC_A2_STATS
    unsigned long long st_full = 0, st_kill = 0, st_removed = 0, st_nunf = 0;
#endif
    // finished-cluster check (cc.cpp:837-974) and publish bookkeeping (cc.cpp:1035-1092) of one column, exact tree state
    bool killed = false; // the last finished-cluster check retired trees
    auto finish_and_publish = [&](const long long gc, const double min_az)
    {
        killed = false;
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
                atomicMax(&T.a_max[j], T.last[i] + 1);
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
                        if (g.mirror_fields)
                        {
                            // final per-tree values of Point (cc.cpp:666-671) for the host mirror; unfinished trees are persisted at the end
                            p.t_fin[cell] = __longlong_as_double((long long) T.fin[i]);
                            p.t_pts[cell] = T.pts[i];
                            p.t_width[cell] = (unsigned) (T.last[i] - T.gcol[i] + 1);
                        }
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
                    T.ring_rel[pos & (TREE_SLOTS - 1)] = gc + WIN_COLS + G; // (+ G: the group-wide verification marks new ids early)
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
            killed = removed > 0;
#ifdef CC_A2_STATS
            st_full++;
            st_kill += removed > 0;
            st_removed += removed;
#endif
            wave_lds_fence();
        }
        last_min_az = min_az;
        if (M_c < first_unpub)
        {
            err = CC_ERR_BOOKKEEPING;
            err_a = M_c;
            err_b = first_unpub;
            return;
        }
        const long long old_unpub = first_unpub;
        first_unpub = M_c;
        ring_start = first_unpub - NC > 0 ? first_unpub - NC : 0;
        emit(CC_EV_PUBLISH_COLUMNS, old_unpub, first_unpub - 1, 0, 0, gc);
        cells_published += (unsigned long long) (first_unpub - old_unpub) * (unsigned long long) R;
    };

#ifdef CC_A2_STATS
    unsigned long long st_ph[5] = {0, 0, 0, 0, 0}, st_sub = 0, st_check = 0, st_live = 0, st_wait = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_pass = 0;
#endif
    long long gc = col_begin; // first column of the current group
    int lc0 = (int) (col_begin % RC);
    load_group(gc, lc0);
    while (gc < col_end && err == 0 && !to_global)
    {
        const int gcount = (int) (col_end - gc < G ? col_end - gc : G);
        const double v_minaz = q_minaz, v_newfin = q_newfin;
        const int v_info = q_info;
        {
            int lcn = lc0 + gcount;
            lcn = lcn >= RC ? lcn - RC : lcn;
            load_group(gc + gcount, lcn); // prefetch: nothing below depends on it
        }

        int u0 = 0;            // first column of the group not yet processed
        bool ids_stale = true; // ids of the columns >= u0 have to be (re)read from the ring
        bool verify = true;    // ... and checked against the tree state (again after trees were finished)
        bool rewalk = true;    // the scalar walk has to be redone (false after a finished-cluster check that retired nothing: only L moved)
        unsigned badmask = 0;
        int v_abad = 0;
        // results of the scalar walk, lane u = column gc + u; they stay valid across a check that retires nothing
        int w_cnt = 0, w_flags = 0, w_maxd = 0, w_nafter = 0, w_nbefore = 0;
        double w_L = 0., w_azprev = 0.;
        long long gcu_l = 0, w_M = 0, w_Mbefore = 0, w_Mc = 0, w_fub = 0;
        bool w_alias = false;
        unsigned long long m_global = 0, m_live = 0, m_check = 0;
        while (u0 < gcount && err == 0 && !to_global)
        {
#ifdef CC_A2_STATS
            unsigned long long tq = __builtin_amdgcn_s_memtime();
#define A2_PH(i)                                                     \
    {                                                                \
        const unsigned long long _n = __builtin_amdgcn_s_memtime();  \
        st_ph[i] += _n - tq;                                         \
        tq = _n;                                                     \
    }
#else
#define A2_PH(i)
#endif
            if (ids_stale)
            {
                wait_a(gc + gcount);
                if (err)
                    break;
                if (lane < G)
                    v_abad = T.info_bad[(int) ((gc + lane) & (A2_INFO - 1))];
                ids_stale = false;
                verify = true;
            }
            A2_PH(0)
            // ---- what wave A assumed: every tree joined from an earlier column is still unfinished (cc.cpp:658). All columns of the
            // group at once: the ids of the trees that start in these columns are marked unfinished first (a freed id stays in
            // quarantine for WIN_COLS + G columns, so no entry of the group can still name its previous tree), then one gather of the
            // flags for every entry of the group — two LDS round trips per group instead of three per column. -------------------------
            if (verify)
            {
                verify = false;
                rewalk = true;
                badmask = 0;
                int q_par[G][RPL], q_e[G][RPL];
#pragma unroll
                for (int u = 0; u < G; u++)
                {
                    const bool on = u >= u0 && u < gcount;
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                    {
                        const int row = k * 64 + lane;
                        const int rr = row < R ? row : 0;
                        const int a = st_parent[(int) ((gc + u) & (A2_STAGE - 1)) * R + rr];
                        const int b = s_win[(int) ((gc + u) & (WIN2_COLS - 1)) * R + rr];
                        q_par[u][k] = (on && row < R) ? a : -2;
                        q_e[u][k] = (on && row < R) ? b : -1;
                    }
                }
#pragma unroll
                for (int u = 0; u < G; u++)
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                        if (q_par[u][k] == -1 && q_e[u][k] >= 0)
                            T.alive[q_e[u][k] & A2_IDMASK] = 1;
                wave_lds_fence();
                unsigned char q_al[G][RPL];
#pragma unroll
                for (int u = 0; u < G; u++)
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                        q_al[u][k] = T.alive[q_e[u][k] >= 0 ? (q_e[u][k] & A2_IDMASK) : 0];
#pragma unroll
                for (int u = 0; u < G; u++)
                {
                    bool bad = false;
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                        bad |= q_par[u][k] >= 0 && (q_e[u][k] < 0 || (!(q_e[u][k] & A2_FRESH) && !q_al[u][k]));
                    if (__any(bad))
                        badmask |= 1u << u;
                }
            }
            A2_PH(1)
            // ---- walk over the columns, one lane per column (lane u = column gc + u): bookkeeping as if no column needed the
            // exact tree state, then the first column that does (the "cut") bounds the batch -----------------------------------
            enum
            {
                CUT_NONE = 0,
                CUT_CHECK = 1, // finished-cluster check may let something through: needs the tree state after this column
                CUT_LIVE = 2,  // the column's static scan result may differ from the live scan: exact serial routine
                CUT_GLOBAL = 3
            };
            const int wu = lane;
            const bool inr = wu >= u0 && wu < gcount;
            const double inf = 1.7976931348623157e308;
            if (rewalk)
            {
                rewalk = false;
                w_cnt = inr ? (v_info & 0xff) : 0;
                w_flags = (v_info >> 8) & 0xff;
                w_maxd = (v_info >> 16) & 0xff;
                const int w_reach = g.mirror_fields ? (v_info >> 24) & 0x7f : 0; // (mirror mode) deepest column any scan of the column looked at
                int ps = w_cnt; // inclusive prefix sums / minima over the columns u0 .. u
                ps += dpp_shr_i32<1>(ps, 0);
                ps += dpp_shr_i32<2>(ps, 0);
                ps += dpp_shr_i32<4>(ps, 0);
                if (G > 8)
                    ps += dpp_shr_i32<8>(ps, 0);
                double pm = (inr && w_cnt > 0) ? v_newfin : inf;
                {
                    double o = dpp_shr_f64<1>(pm, inf);
                    pm = o < pm ? o : pm;
                    o = dpp_shr_f64<2>(pm, inf);
                    pm = o < pm ? o : pm;
                    o = dpp_shr_f64<4>(pm, inf);
                    pm = o < pm ? o : pm;
                    if (G > 8)
                    {
                        o = dpp_shr_f64<8>(pm, inf);
                        pm = o < pm ? o : pm;
                    }
                }
                w_nafter = n_unf + ps;
                w_nbefore = w_nafter - w_cnt;
                w_L = pm < L ? pm : L;
                gcu_l = gc + wu;
                // the oldest root column: set by the first new tree while there is none (cc.cpp:1035-1050 keeps the minimum)
                int firstnew = 64;
                if (n_unf == 0)
                {
                    const unsigned long long nm = __ballot(inr && w_cnt > 0);
                    firstnew = nm ? (int) __ffsll((long long) nm) - 1 : 64;
                }
                w_M = (n_unf == 0 && wu >= firstnew) ? gc + firstnew : M;
                w_Mbefore = (n_unf == 0 && wu > firstnew) ? gc + firstnew : M;
                w_Mc = w_nafter == 0 ? gcu_l + 1 : w_M; // first unpublished column after this column, nothing finishing
                w_fub = dpp_shr_i64<1>(w_Mc, first_unpub);
                w_fub = wu == u0 ? first_unpub : w_fub;
                w_azprev = dpp_shr_f64<1>(v_minaz, last_min_az);
                w_azprev = wu == u0 ? last_min_az : w_azprev;
                const bool w_badbit = (badmask >> wu) & 1u;
                const bool c_global = inr && (w_nafter > tree_limit || (v_abad & 3) == 2);
                const bool c_live = inr && (w_badbit || (w_flags & 1) || (v_abad & 3) == 1 ||
                                            (w_nbefore > 0 && (uint32_t) (gcu_l - w_Mbefore + 1) > (uint32_t) NC) // cc.cpp:657
                                            || gcu_l - w_maxd < w_fub                                              // cc.cpp:762-763
                                            || gcu_l - w_reach < w_fub); // static visit counts (cc.cpp:725) need the whole window
                w_alias = w_nafter > 0 && v_minaz == w_azprev;
                const bool c_check = inr && w_nafter > 0 && !w_alias && ((gcu_l + 1 - w_M) >= NC || v_minaz >= w_L);
                m_global = __ballot(c_global);
                m_live = __ballot(c_live);
                m_check = __ballot(c_check);
            }
            else
            {
                // A finished-cluster check ran with the exact tree state and retired nothing: tree count, oldest root, first unpublished
                // column of the later columns are what the walk said; only the bound L was refreshed. The prefix minimum over the new
                // roots' finished_at restarts behind the checked column (the earlier ones are part of L now).
                double pm = (inr && w_cnt > 0) ? v_newfin : inf;
                {
                    double o = dpp_shr_f64<1>(pm, inf);
                    pm = o < pm ? o : pm;
                    o = dpp_shr_f64<2>(pm, inf);
                    pm = o < pm ? o : pm;
                    o = dpp_shr_f64<4>(pm, inf);
                    pm = o < pm ? o : pm;
                    if (G > 8)
                    {
                        o = dpp_shr_f64<8>(pm, inf);
                        pm = o < pm ? o : pm;
                    }
                }
                w_L = pm < L ? pm : L;
                const bool c_check = inr && w_nafter > 0 && !w_alias && ((gcu_l + 1 - w_M) >= NC || v_minaz >= w_L);
                m_check = __ballot(c_check);
                const unsigned long long keep = ~((1ull << u0) - 1ull);
                m_global &= keep;
                m_live &= keep;
            }
            const unsigned long long m_cut = m_global | m_live | m_check;
            const int ucut = m_cut ? (int) __ffsll((long long) m_cut) - 1 : gcount;
            int cut = CUT_NONE;
            if (m_cut)
                cut = ((m_global >> ucut) & 1ull) ? CUT_GLOBAL : (((m_live >> ucut) & 1ull) ? CUT_LIVE : CUT_CHECK);
            const int u1 = ucut + (cut == CUT_CHECK ? 1 : 0); // columns [u0, u1) are applied as one batch
            const int ucomp = ucut;                            // columns [u0, ucomp) are complete (checked + published)
            const bool w_done = wu >= u0 && wu < ucomp;
            {
                const unsigned long long m_err = __ballot(w_done && w_Mc < w_fub);
                if (m_err)
                {
                    const int ue = (int) __ffsll((long long) m_err) - 1;
                    err = CC_ERR_BOOKKEEPING;
                    err_a = lane_i64(w_Mc, ue);
                    err_b = lane_i64(w_fub, ue);
                    break;
                }
            }
            if (g.record_events)
            {
                // per complete column: ground-column event, publish event (cc.cpp:618-620, 1087-1089)
                const int idx = n_events + 2 * (wu - u0);
                if (w_done && idx + 1 < g.event_capacity + 1)
                {
                    cc_event e;
                    e.stream = s;
                    e.c = 0;
                    e.d = 0;
                    e.column = gcu_l;
                    if (idx < g.event_capacity)
                    {
                        e.type = CC_EV_GROUND_COLUMN;
                        e.a = gcu_l;
                        e.b = gcu_l;
                        p.events[idx] = e;
                    }
                    if (idx + 1 < g.event_capacity)
                    {
                        e.type = CC_EV_PUBLISH_COLUMNS;
                        e.a = w_fub;
                        e.b = w_Mc - 1;
                        p.events[idx + 1] = e;
                    }
                }
                n_events += 2 * (ucomp - u0);
            }
            if (ucomp > u0)
            {
                const long long fu_new = lane_i64(w_Mc, ucomp - 1);
                cells_published += (unsigned long long) (fu_new - first_unpub) * (unsigned long long) R;
                first_unpub = fu_new;
                ring_start = first_unpub - NC > 0 ? first_unpub - NC : 0;
                last_min_az = lane_f64(v_minaz, ucomp - 1);
                alias_rounds += (unsigned long long) __popcll(__ballot(w_done && w_alias));
            }
            if (u1 > u0)
            {
                n_unf = lane_i32(w_nafter, u1 - 1);
                L = lane_f64(w_L, u1 - 1);
                M = lane_i64(w_M, u1 - 1);
            }
            if (cut == CUT_CHECK)
                emit(CC_EV_GROUND_COLUMN, gc + ucut, gc + ucut, 0, 0, gc + ucut);
            A2_PH(2)

            // ---- the batch [u0, u1), column by column: new trees (list order = column, then row), then the point and link
            // updates. Point updates are run-length aggregated per row: consecutive columns of a row mostly join the same tree, so
            // a lane keeps (tree, points, max finished_at, last column) in registers and touches the tree state only when its
            // tree changes and at the end of the batch. The column's inputs are read one column ahead (no LDS wait in the loop); the
            // root plane was written by wave A. ---------------------------------------------------------------------------------
            if (u1 > u0)
            {
                int cur[RPL];
                unsigned rcnt[RPL];
                unsigned long long rfin[RPL];
                long long rlast[RPL];
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    cur[k] = -1;
                    rcnt[k] = 0;
                    rfin[k] = 0;
                    rlast[k] = 0;
                }
                auto flush = [&](int k)
                {
                    const int i = cur[k];
                    const int rep = lds_find(T.uf, i);
                    atomicMax(&T.last[i], rlast[k]);
                    atomicMax(&T.fin[i], rfin[k]);
                    atomicMax(&T.c_fin[rep], rfin[k]);
                    atomicAdd(&T.pts[i], rcnt[k]);
                };
                int n_par[RPL], n_e[RPL];
                double n_fc[RPL];
                auto load_col = [&](int u)
                {
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                    {
                        const int row = k * 64 + lane;
                        const int rr = row < R ? row : 0;
                        const int so = (int) ((gc + u) & (A2_STAGE - 1)) * R + rr;
                        const int a = st_parent[so];
                        const double f = st_fin[so];
                        const int b = s_win[(int) ((gc + u) & (WIN2_COLS - 1)) * R + rr];
                        n_par[k] = row < R ? a : -2;
                        n_fc[k] = f;
                        n_e[k] = row < R ? b : -1;
                    }
                };
                load_col(u0);
                int lcu = lc0 + u0;
                lcu = lcu >= RC ? lcu - RC : lcu;
                for (int u = u0; u < u1; u++, lcu = (lcu + 1 == RC ? 0 : lcu + 1))
                {
                    const long long gcu = gc + u;
                    const int info = lane_i32(v_info, u);
                    const bool has_new = (info & 0xff) != 0, has_links = (lane_i32(v_abad, u) >> 4) & 1; // links to other trees
                    int par[RPL], e[RPL];
                    double fc[RPL];
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                    {
                        par[k] = n_par[k];
                        e[k] = n_e[k];
                        fc[k] = n_fc[k];
                    }
                    if (u + 1 < u1)
                        load_col(u + 1);
                    if (has_new)
                    {
                        const int nb = lane_i32(w_nbefore, u);
                        int cnt = 0;
#pragma unroll
                        for (int k = 0; k < RPL; k++)
                        {
                            const int row = k * 64 + lane;
                            const bool is_new = par[k] == -1;
                            const unsigned long long mask = __ballot(is_new);
                            if (is_new)
                            {
                                const int i = e[k] & A2_IDMASK;
                                T.cell[i] = lcu * R + row;
                                T.gcol[i] = gcu;
                                T.fin[i] = (unsigned long long) __double_as_longlong(fc[k]);
                                T.last[i] = gcu;
                                T.pts[i] = 1;
                                T.uf[i] = i;
                                T.c_fin[i] = (unsigned long long) __double_as_longlong(fc[k]);
                                T.alist[nb + cnt + __popcll(mask & lanes_below())] = (short) i;
                                T.alive[i] = 1;
                            }
                            cnt += __popcll(mask);
                        }
                        wave_lds_fence();
                    }
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                    {
                        const int i = e[k] & A2_IDMASK;
                        if (e[k] >= 0 && i != cur[k])
                        {
                            if (rcnt[k] > 0)
                                flush(k);
                            cur[k] = i;
                            rcnt[k] = 0;
                            rfin[k] = 0;
                        }
                        if (par[k] >= 0)
                        {
                            const unsigned long long fb = (unsigned long long) __double_as_longlong(fc[k]);
                            rcnt[k]++;
                            rfin[k] = fb > rfin[k] ? fb : rfin[k];
                            rlast[k] = gcu;
                        }
                    }
                    if (has_links)
                    {
#pragma unroll
                        for (int k = 0; k < RPL; k++)
                        {
                            const int row = k * 64 + lane;
                            const int wcu = (int) (gcu & (WIN2_COLS - 1));
                            const int nlk = (par[k] >= 0 && row < R) ? (int) p.sc_nlinks[lcu * R + row] : 0; // (rare path: straight from HBM)
                            if (nlk > 0)
                            {
                                const int i = e[k] & A2_IDMASK;
                                const unsigned long long lk = p.sc_links[lcu * R + row];
                                for (int j = 0; j < nlk; j++) // (rare path since wave A filters the columns: small, not fast)
                                {
                                    const int code = (int) ((lk >> (16 * j)) & 0xffff);
                                    const int v = s_win[((wcu - (code >> 8)) & (WIN2_COLS - 1)) * R + (code & 0xff)];
                                    const int vv = v & A2_IDMASK;
                                    if (v >= 0 && vv != i && T.alive[vv])
                                    {
                                        log_link(g, st, p.link_log, T.cell[i], T.cell[vv]);
                                        lds_union(T.uf, T.c_fin, i, vv);
                                    }
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < RPL; k++)
                    if (rcnt[k] > 0)
                        flush(k);
                wave_lds_fence();
            }

            A2_PH(3)
            // ---- the cut column ----------------------------------------------------------------------------------------------
#ifdef CC_A2_STATS
            st_nunf += n_unf;
            st_sub++;
            st_check += cut == CUT_CHECK;
            st_live += cut == CUT_LIVE;
#endif
            int check_u = -1; // column whose finished-cluster check runs with the exact tree state (one call site: code size)
            if (cut == CUT_CHECK)
            {
                check_u = u1 - 1;
                u0 = u1;
            }
            else if (cut == CUT_LIVE)
            {
                const int u = u1;
                const long long gcu = gc + u;
                int lcu = lc0 + u;
                lcu = lcu >= RC ? lcu - RC : lcu;
                emit(CC_EV_GROUND_COLUMN, gcu, gcu, 0, 0, gcu);
                serial_cols++;
                wait_a(gcu + 1); // the hand-off record of this column (ring head before it)
                if (err)
                    break;
                const int info_head = uniform_i32(lds_ld(&T.info_head[(int) (gcu & (A2_INFO - 1))]));
                park_a();
                if (err)
                    break;
                if (lane == 0)
                {
                    int nn = n_unf, e = 0, hd = info_head;
                    double LL = L;
                    long long MM = M;
                    assoc_column_live2<RPL>(c, cfg, g, T, s_win, lcu, gcu, (int) (first_unpub % RC), nn, LL, MM, hd, e, st); // (64-bit modulo: rare path)
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
                    err = CC_ERR_CAPACITY; // the live replay ran out of tree ids mid-column: the column cannot be rolled back
                    err_a = n_unf;
                }
                L = uniform_f64(T.bcast_d[0]);
                M = uniform_i64(T.bcast_l[0]);
                wave_lds_fence();
                if (err)
                    break;
                resume_a(gcu + 1, hd);
                check_u = u;
                u0 = u1 + 1;
                ids_stale = true; // wave A resolves the rest of the group again
            }
            else if (cut == CUT_GLOBAL)
            {
                to_global = true; // continue this stream with the global-memory kernel, starting at this column
                gc += u1;
                break;
            }
            else
                u0 = gcount;
            if (check_u >= 0)
            {
                finish_and_publish(gc + check_u, lane_f64(v_minaz, check_u));
                verify |= killed;
            }
            if (lane == 0)
                lds_st(&T.b_done, gc + u0);
            A2_PH(4)
        }
        if (to_global || err)
            break;
        gc += gcount;
        lc0 += gcount;
        lc0 = lc0 >= RC ? lc0 - RC : lc0;
    }
    if (lane == 0)
        lds_st(&T.cmd, (int) A2_EXIT);
#ifdef CC_A2_STATS
    if (lane == 0)
    {
        st->dbg[8] += st_sub;
        st->dbg[9] += st_check;
        st->dbg[10] += __builtin_amdgcn_s_memtime() - st_t0;
        st->dbg[11] += st_live;
        st->dbg[12] += (unsigned long long) (gc - col_begin);
        st->dbg[13] += st_wait_g;
        st->dbg[14] += st_full;
        st->dbg[15] += st_kill;
        st->dbg[5] += st_removed;
        st->dbg[6] += st_nunf;
        for (int i = 0; i < 5; i++)
            st->dbg[i] += st_ph[i];
    }
#endif

    // ---- persist the tree state back to the global planes: list order = creation order -----------------------------------------
    {
        wave_lds_fence();
        for (int r = lane; r < n_unf; r += 64)
        {
            const int i = T.alist[r];
            const int cell = T.cell[i];
            p.ulist[r] = cell;
            p.t_pos[cell] = r;
            p.t_fin[cell] = __longlong_as_double((long long) T.fin[i]);
            p.t_width[cell] = (unsigned) (T.last[i] - T.gcol[i] + 1);
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
