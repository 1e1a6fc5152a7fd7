// cc_k_assoc_global.h — the window scan of one point (scan_point), the global-memory association fallback (associate_stream / k_associate), the column epilogue of the window scan.
// (part of cc_kernels.h: included there, in order, inside namespace cck)
#pragma once

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
