// What the cooperating-wavefront association kernels share (included by cc_kernels.h inside namespace cck, in front of cc_assoc3.h): the LDS tree
// table with stable ids, the per-cluster finish prediction, the exact serial replay of one column (assoc_column_live2) and the DPP shift helpers.
// They were written for k_assoc2, the two-wavefront kernel of round 1 (front wavefront resolves tree ids ahead, back wavefront applies them);
// k_assoc3 took its place in round 2 and the batch-parallel k_assocb went in front of that in round 3. Round 4 retired k_assoc2 itself
// (1 200 lines that were neither the default nor a fallback of anything): the serial fallbacks are k_assoc3 (LDS) and k_associate (global memory).
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
