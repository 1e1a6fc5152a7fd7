// cc_k_assoc_lds.h — k_assoc_lds: the one-wavefront serial association (cluster_point_trees_every_nth_column != 1, option assoc_waves = 1).
// (part of cc_kernels.h: included there, in order, inside namespace cck)
#pragma once

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
        const CazBase cbx = caz_base_of_column(gcx, g.num_columns);
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
                nx_fin[k] = cell_fin_of(g, cfg, p, ci, cbx);
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
