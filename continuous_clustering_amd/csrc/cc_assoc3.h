// Three-wave association kernel (included by cc_kernels.h inside namespace cck, after cc_assoc_shared.h whose structures and helpers it uses).
//
// k_assoc3 is k_assoc2 with the per-point work taken off the back wave. k_assoc2's back wave needs ~2 300 clocks per column, of which
// ~1 000 go into looking at every point of the column twice (liveness check, run-length aggregation) although a column touches one to
// three trees. Here a third wavefront ("records wave") sits between the two: it turns each resolved column into a few RECORDS — the new
// roots in row order; per tree that receives points their number and the largest finished_at contribution — and writes the tree-root
// plane. The back wave then checks and applies whole groups of columns with one lane per record (two LDS round trips per group), keeps its
// scalar walk, the exact finished-cluster check and the publish bookkeeping. Columns whose records do not fit (more than A3_REC trees or
// A3_BIRTH new roots in one column) take the exact serial replay like every other exception. Same results as k_assoc2, bit for bit
// (tests: every parity case runs with it).
#pragma once

constexpr int A3_REC = 8;   // records (trees receiving points) per column
constexpr int A3_THREADS = 256; // four wavefronts: A (resolve ids), B (apply + finish), R (records + roots), L (links). Launched with 192
                                // threads the kernel runs without wave L and wave A looks at the links itself (1.74 instead of 1.55 ms per
                                // 2200 columns): one wavefront less per stream for the throughput kernels it shares the GPU with, which
                                // only pays when a launch has more streams than CUs (cc_engine.hip: CC_LWAVE_MAX_STREAMS)
constexpr int A3_PSTAGE = 8; // columns of per-point inputs staged between wave A and wave R (power of two)
constexpr int A3_BIRTH = 8; // new roots per column kept inline (must equal A3_REC: one lane per (column, slot))

// One stream's batch (or what k_assocb left of it). The kernel below calls it for the streams of its block: one block per stream when the serial
// kernel is what associates (assoc_batch off, or k_assocb had to stop lately), a handful of blocks that sweep over all streams — and find nothing to do
// — when it is only the safety net behind k_assocb: 256 blocks of 256 threads and 45 KB of LDS each took 0.2 - 0.5 ms of chain time to be PLACED next
// to the other chains' kernels, just to return.
template<int RPL>
__device__ __forceinline__ void assoc3_stream(const Geometry& g, const cc_config& cfg, const Planes& P, StreamState* states, const int s, const int slot,
                                              const int limited)
{
    constexpr int G = RPL == 1 ? 8 : 4; // columns wave B handles per pass
    constexpr int A2_LEAD = a2_lead(RPL), A2_STAGE = a2_stage(RPL);
    static_assert(WIN_COLS + A2_LEAD + 1 <= WIN2_COLS && A2_LEAD + G <= A2_STAGE && A2_LEAD < A2_INFO, "ring sizes");
    const int lane = lane_id();
    const int wave = uniform_i32((int) (threadIdx.x >> 6));
    const bool lwave = blockDim.x > 192; // wave L exists
    const int nthreads = (int) blockDim.x;
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
    // (only wave R reads them, right behind wave A: a ring of A3_PSTAGE columns, and wave A never runs further ahead of wave R than that.
    // Staging all A2_STAGE columns cost 15 KB more LDS per block, which the other chains' blocks on the CU could not use: 32 KB more made the
    // 256-stream step 8 % slower.)
    __shared__ double st_fin[A3_PSTAGE * WAVE * RPL];
    __shared__ short st_parent[A3_PSTAGE * WAVE * RPL];
    // per column between the records wave and wave B: what the column does to the tree state, as a few records instead of 64-128 points
    __shared__ short rc_id[A2_STAGE][A3_REC];                // trees that receive points of the column ...
    __shared__ unsigned short rc_cnt[A2_STAGE][A3_REC];      // ... how many ...
    __shared__ unsigned long long rc_fin[A2_STAGE][A3_REC];  // ... and the largest finished_at contribution (bits of a non-negative double)
    __shared__ short bt_id[A2_STAGE][A3_BIRTH];              // new roots of the column in row order: tree id,
    __shared__ unsigned short bt_row[A2_STAGE][A3_BIRTH];    // row,
    __shared__ unsigned long long bt_fin[A2_STAGE][A3_BIRTH]; // finished_at
    __shared__ unsigned char rc_n[A2_STAGE], bt_n[A2_STAGE];  // counts; 255 = more than fit (the column is replayed exactly)
    __shared__ long long r_done;                              // columns < r_done have their records
    __shared__ int r_parked;
    // the links wave: per column whether a link candidate (an accepted candidate after the first, cc.cpp:693-694) leads to another tree
    __shared__ long long l_done;                              // columns < l_done have their flag
    __shared__ int l_parked;
    __shared__ unsigned char l_foreign[A2_INFO];

    // limited: only up to where the batch-parallel kernel asked (the group it could not take); it is launched again behind this launch
    const long long col_begin = st->batch[slot].acp_next, first_column = st->first_column;
    const long long col_end = (limited && st->serial_until < st->batch[slot].seg_end) ? st->serial_until : st->batch[slot].seg_end;
    if (col_begin >= col_end)
        return;
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
    for (int i = threadIdx.x; i < TREE_SLOTS; i += nthreads)
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
        r_done = col_begin;
        r_parked = 0;
        l_done = col_begin;
        l_parked = 0;
        for (int i = 0; i < A2_INFO; i++)
            l_foreign[i] = 0;
        T.a_done = col_begin;
        T.b_done = col_begin;
        T.restart_col = col_begin;
        T.cmd = A2_RUN;
        T.a_parked = 0;
        T.head = 0;
        T.tail = TREE_SLOTS - n_unf0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_unf0; i += nthreads)
        atomicMax(&T.c_fin[lds_find(T.uf, i)], T.fin[i]);
    {
        // ring of tree ids for the WIN2_COLS columns before col_begin (only the last WIN_COLS can be looked at): two dependent
        // gathers per cell (root plane, then the tree planes at the root), 8 cells at a time
        constexpr int B = 8;
        for (int i0 = threadIdx.x; i0 < WIN2_COLS * R; i0 += nthreads * B)
        {
            int rr[B];
#pragma unroll
            for (int u = 0; u < B; u++)
            {
                const int i = i0 + u * nthreads;
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
                const int i = i0 + u * nthreads;
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
#ifdef CC_A2_STATS
        unsigned long long st_acyc = 0, st_acols = 0; // busy cycles (s_memtime) and columns of this wave
#endif
        int head = 0;
        long long gcA = col_begin;
        int lc = (int) (col_begin % RC);
        long long b_seen = col_begin, r_seen = col_begin;
        int nx_term[RPL], nx_info = 0, nx_par[RPL];
        int nx_nl[RPL];                // (link candidates: only without wave L)
        unsigned long long nx_link[RPL];
        double nx_fin[RPL];
        auto load_a = [&](long long gcx, int lcx)
        {
            const CazBase cbx = caz_base_of_column(gcx, g.num_columns);
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
                    nx_fin[k] = cell_fin_of(g, cfg, p, lcx * R + row, cbx);
                    nx_term[k] = p.sc_term[lcx * R + row];
                    if (!lwave)
                    {
                        nx_nl[k] = p.sc_nlinks[lcx * R + row];
                        nx_link[k] = p.sc_links[lcx * R + row]; // (stale where the point has no links: never looked at)
                    }
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
            const bool idle = wait_park || gcA >= col_end || gcA - b_seen >= A2_LEAD || gcA - r_seen >= A3_PSTAGE;
            if (idle || (++poll & 3) == 0)
            {
                const int cmd = uniform_i32(lds_ld(&T.cmd));
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
                    r_seen = gcA;
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
                if (gcA - r_seen >= A3_PSTAGE) // the staging ring is full: wave R has to take a column first
                {
                    r_seen = uniform_i64(lds_ld(&r_done));
                    if (gcA - r_seen >= A3_PSTAGE)
                    {
                        __builtin_amdgcn_s_sleep(1);
                        continue;
                    }
                }
            }
#ifdef CC_A2_STATS
            const unsigned long long st_ta = __builtin_amdgcn_s_memtime();
#endif
            int term[RPL], parc[RPL], nlk[RPL];
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
                    const int o = (int) (gcA & (A3_PSTAGE - 1)) * R + row;
                    st_parent[o] = (short) nx_par[k];
                    st_fin[o] = nx_fin[k];
                }
            }
            const int cnt_new = uniform_i32(nx_info) & 0xff;
            const bool col_links = !lwave && ((uniform_i32(nx_info) >> 8) & 2);
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
            // Links (further accepted candidates) only matter where they lead to another tree, which is rare (two trees of one object
            // meeting). Without wave L this wave looks the targets up and tells wave B whether the column has any.
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
#ifdef CC_A2_STATS
            st_acyc += __builtin_amdgcn_s_memtime() - st_ta;
            st_acols++;
#endif
        }
#ifdef CC_A2_STATS
        if (lane == 0)
        {
            atomicAdd((unsigned long long*) &st->dbg[6], st_acyc);
            atomicAdd((unsigned long long*) &st->dbg[7], st_acols);
        }
#endif
        return;
    }

    if (wave == 2)
    {
        // ============================================================================================= wave R: records + tree roots
        // Behind wave A, ahead of wave B. Per column (lanes = rows): the new roots in row order; per tree that receives points the
        // number of points and the largest finished_at contribution (a handful of wave-wide reductions: a column touches 1-3 trees);
        // Point::tree_root_ of every cell (cc.cpp:661,814): a new root's lane files its cell under the tree id, every lane gathers its
        // tree's root cell and writes the root plane. Wave B then never looks at a point: it applies these records.
#ifdef CC_A2_STATS
        unsigned long long st_rcyc = 0;
#endif
        long long gcR = col_begin, a_seen = col_begin;
        int lcR = (int) (col_begin % RC);
        int poll = 0;
        while (true)
        {
            if (gcR >= col_end || gcR >= a_seen || (++poll & 7) == 0)
            {
                const int cmd = uniform_i32(lds_ld(&T.cmd));
                if (cmd == A2_EXIT)
                    break;
                if (cmd == A2_PARK)
                {
                    // (the tree-root stores of the columns this wave handled must have landed before wave B replays one of them)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0)
                        lds_st(&r_parked, 1);
                    while (uniform_i32(lds_ld(&T.cmd)) == A2_PARK)
                        __builtin_amdgcn_s_sleep(1);
                    if (uniform_i32(lds_ld(&T.cmd)) == A2_EXIT)
                        break;
                    wave_lds_fence();
                    gcR = uniform_i64(lds_ld(&T.restart_col));
                    lcR = (int) (gcR % RC);
                    a_seen = gcR;
                    continue;
                }
                if (gcR >= col_end)
                {
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                if (gcR >= a_seen)
                {
                    a_seen = uniform_i64(lds_ld(&T.a_done));
                    if (gcR >= a_seen)
                    {
                        __builtin_amdgcn_s_sleep(1);
                        continue;
                    }
                    wave_lds_fence(); // ring entries and staged inputs are read after the flag
                }
            }
#ifdef CC_A2_STATS
            const unsigned long long st_tr = __builtin_amdgcn_s_memtime();
#endif
            const int sc = (int) (gcR & (A2_STAGE - 1));
            const int abad = uniform_i32(lds_ld(&T.info_bad[(int) (gcR & (A2_INFO - 1))]));
            if ((abad & 3) == 0)
            {
                int par[RPL], e[RPL];
                unsigned long long fb[RPL];
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    const int rr = row < R ? row : 0;
                    const int a = st_parent[(int) (gcR & (A3_PSTAGE - 1)) * R + rr];
                    const double f = st_fin[(int) (gcR & (A3_PSTAGE - 1)) * R + rr];
                    const int b = s_win[(int) (gcR & (WIN2_COLS - 1)) * R + rr];
                    par[k] = row < R ? a : -2;
                    e[k] = row < R ? b : -1;
                    fb[k] = (unsigned long long) __double_as_longlong(f);
                }
                // new roots, row order
                int nb = 0;
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const bool is_new = par[k] == -1 && e[k] >= 0;
                    const unsigned long long m = __ballot(is_new);
                    if (is_new)
                    {
                        const int pos = nb + __popcll(m & lanes_below());
                        if (pos < A3_BIRTH)
                        {
                            bt_id[sc][pos] = (short) (e[k] & A2_IDMASK);
                            bt_row[sc][pos] = (unsigned short) (k * 64 + lane);
                            bt_fin[sc][pos] = fb[k];
                        }
                        T.cell[e[k] & A2_IDMASK] = lcR * R + k * 64 + lane;
                    }
                    nb += __popcll(m);
                }
                // per tree that receives points: count and maximum
                int nr = 0;
                unsigned long long todo[RPL];
#pragma unroll
                for (int k = 0; k < RPL; k++)
                    todo[k] = __ballot(par[k] >= 0 && e[k] >= 0);
                bool orphan = false; // a point with a parent but without a tree id: the exact routine decides
#pragma unroll
                for (int k = 0; k < RPL; k++)
                    orphan |= par[k] >= 0 && e[k] < 0;
                while (true)
                {
                    int src_k = -1;
#pragma unroll
                    for (int k = RPL - 1; k >= 0; k--)
                        if (todo[k])
                            src_k = k;
                    if (src_k < 0)
                        break;
                    int X = 0;
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                        if (k == src_k)
                            X = __builtin_amdgcn_readlane(e[k], (int) __ffsll((long long) todo[k]) - 1) & A2_IDMASK;
                    int cnt = 0;
                    unsigned long long mx = 0;
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                    {
                        const bool mine = par[k] >= 0 && e[k] >= 0 && (e[k] & A2_IDMASK) == X;
                        const unsigned long long m = __ballot(mine);
                        cnt += __popcll(m);
                        todo[k] &= ~m;
                        const unsigned long long v = mine ? fb[k] : 0ull;
                        mx = v > mx ? v : mx;
                    }
                    mx = wave_max_f64_bits(mx); // (finished_at contributions: non-negative doubles)
                    if (lane == 0 && nr < A3_REC)
                    {
                        rc_id[sc][nr] = (short) X;
                        rc_cnt[sc][nr] = (unsigned short) cnt;
                        rc_fin[sc][nr] = mx;
                    }
                    nr++;
                }
                const bool any_orphan = __any(orphan);
                if (lane == 0)
                {
                    rc_n[sc] = (unsigned char) ((nr > A3_REC || any_orphan) ? 255 : nr);
                    bt_n[sc] = (unsigned char) (nb > A3_BIRTH ? 255 : nb);
                }
                // tree roots (handing them to the links wave, which then has to run behind this one, made the chain 3 % slower)
                wave_lds_fence();
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    const int cell = T.cell[e[k] >= 0 ? (e[k] & A2_IDMASK) : 0];
                    if (row < R)
                        p.root[lcR * R + row] = e[k] >= 0 ? cell : -1;
                }
            }
            else if (lane == 0)
            {
                rc_n[sc] = 0;
                bt_n[sc] = 0;
            }
            wave_lds_fence();
            if (lane == 0)
                lds_st(&r_done, gcR + 1);
            gcR++;
            lcR = lcR + 1 == RC ? 0 : lcR + 1;
#ifdef CC_A2_STATS
            st_rcyc += __builtin_amdgcn_s_memtime() - st_tr;
#endif
        }
#ifdef CC_A2_STATS
        if (lane == 0)
            atomicAdd((unsigned long long*) &st->dbg[5], st_rcyc);
#endif
        return;
    }

    if (wave == 3)
    {
        // ============================================================================================= wave L: links
        // Behind wave A, ahead of wave B. Links (further accepted candidates, cc.cpp:693-694) only matter where they lead to another
        // tree, which is rare (two trees of one object meeting) — but nine columns of ten have link candidates, and looking their
        // targets up in the id ring was 40 % of wave A's column. This wave does nothing else: it reads the link words itself (two
        // columns ahead), looks the targets up and tells wave B per column whether any of them is foreign.
        long long gcL = col_begin, a_seen = col_begin;
        int lcL = (int) (col_begin % RC);
        int poll = 0;
        // inputs of columns gcL (c_*) and gcL + 1 (n_*)
        int c_nl[RPL], n_nl[RPL], c_info = 0, n_info = 0;
        unsigned long long c_lk[RPL], n_lk[RPL];
        auto load_l = [&](long long gcx, int lcx, int* nl, unsigned long long* lk, int& info)
        {
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                const int row = k * 64 + lane;
                nl[k] = 0;
                lk[k] = 0;
                if (row < R && gcx < col_end)
                {
                    nl[k] = p.sc_nlinks[lcx * R + row];
                    lk[k] = p.sc_links[lcx * R + row]; // (stale where the point has no links: never looked at)
                }
            }
            info = 0;
            if (gcx < col_end)
                info = p.col_info[lcx];
        };
        auto reload = [&]()
        {
            load_l(gcL, lcL, c_nl, c_lk, c_info);
            load_l(gcL + 1, lcL + 1 == RC ? 0 : lcL + 1, n_nl, n_lk, n_info);
        };
        reload();
        while (true)
        {
            if (gcL >= col_end || gcL >= a_seen || (++poll & 7) == 0)
            {
                const int cmd = uniform_i32(lds_ld(&T.cmd));
                if (cmd == A2_EXIT)
                    break;
                if (cmd == A2_PARK)
                {
                    if (lane == 0)
                        lds_st(&l_parked, 1);
                    while (uniform_i32(lds_ld(&T.cmd)) == A2_PARK)
                        __builtin_amdgcn_s_sleep(1);
                    if (uniform_i32(lds_ld(&T.cmd)) == A2_EXIT)
                        break;
                    wave_lds_fence();
                    gcL = uniform_i64(lds_ld(&T.restart_col));
                    lcL = (int) (gcL % RC);
                    a_seen = gcL;
                    reload();
                    continue;
                }
                if (gcL >= col_end)
                {
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                if (gcL >= a_seen)
                {
                    a_seen = uniform_i64(lds_ld(&T.a_done));
                    if (gcL >= a_seen)
                    {
                        __builtin_amdgcn_s_sleep(1);
                        continue;
                    }
                    wave_lds_fence(); // ring entries are read after the flag
                }
            }
            int nlk[RPL];
            unsigned long long lk[RPL];
#pragma unroll
            for (int k = 0; k < RPL; k++)
            {
                nlk[k] = c_nl[k];
                lk[k] = c_lk[k];
                c_nl[k] = n_nl[k];
                c_lk[k] = n_lk[k];
            }
            const bool col_links = (uniform_i32(c_info) >> 8) & 2;
            c_info = n_info;
            {
                int lc2 = lcL + 2;
                lc2 = lc2 >= RC ? lc2 - RC : lc2;
                load_l(gcL + 2, lc2, n_nl, n_lk, n_info); // two columns ahead
            }
            const int abad = uniform_i32(lds_ld(&T.info_bad[(int) (gcL & (A2_INFO - 1))]));
            int foreign = 0;
            if (col_links && (abad & 3) == 0)
            {
                const int wcur = (int) (gcL & (WIN2_COLS - 1));
                bool f = false;
#pragma unroll
                for (int k = 0; k < RPL; k++)
                {
                    const int row = k * 64 + lane;
                    const int e = row < R ? (int) s_win[wcur * R + row] : -1; // the point's tree (>= 0: the point has one)
                    const int mine = e & A2_IDMASK;
                    int v[LINK_SLOTS];
#pragma unroll
                    for (int j = 0; j < LINK_SLOTS; j++)
                    {
                        v[j] = -1;
                        if (e >= 0 && j < nlk[k])
                        {
                            const int code = (int) ((lk[k] >> (16 * j)) & 0xffff);
                            v[j] = s_win[((wcur - (code >> 8)) & (WIN2_COLS - 1)) * R + (code & 0xff)];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < LINK_SLOTS; j++)
                        f |= v[j] >= 0 && (v[j] & A2_IDMASK) != mine;
                }
                foreign = __any(f) ? 1 : 0;
            }
            if (lane == 0)
                l_foreign[(int) (gcL & (A2_INFO - 1))] = (unsigned char) foreign;
            wave_lds_fence();
            if (lane == 0)
                lds_st(&l_done, gcL + 1);
            gcL++;
            lcL = lcL + 1 == RC ? 0 : lcL + 1;
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
        while (uniform_i32(lds_ld(&T.a_parked)) == 0 || uniform_i32(lds_ld(&r_parked)) == 0 || (lwave && uniform_i32(lds_ld(&l_parked)) == 0))
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
            r_done = restart;
            r_parked = 0;
            l_done = restart;
            l_parked = 0;
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
            a_seen = uniform_i64(lds_ld(&r_done)); // (records ready = resolved by wave A and summarised by wave R ...
            if (lwave)
            {
                const long long l_seen = uniform_i64(lds_ld(&l_done)); // ... and the links looked at by wave L)
                a_seen = l_seen < a_seen ? l_seen : a_seen;
            }
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
        int q_id = -1;                 // this lane's record of the group (tree id, -1: none), kept from the verification for the walks
        unsigned long long q_fin = 0;  // ... and its finished_at contribution
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
                    v_abad = T.info_bad[(int) ((gc + lane) & (A2_INFO - 1))] | (l_foreign[(int) ((gc + lane) & (A2_INFO - 1))] ? 16 : 0);
                ids_stale = false;
                verify = true;
            }
            A2_PH(0)
            // ---- what wave A assumed: every tree that receives points is still unfinished (cc.cpp:658). One lane per (column, record):
            // the new roots of the group are marked unfinished first (a freed id stays in quarantine for WIN_COLS + G columns, so no
            // record of the group can still mean its previous tree), then one gather of the flags. A column whose records or new roots
            // did not fit is replayed exactly as well. -------------------------------------------------------------------------------
            const int ru = lane / A3_REC, rk = lane % A3_REC;               // this lane's (column of the group, record / new-root slot)
            const int rsc = (int) ((gc + ru) & (A2_STAGE - 1));
            if (verify)
            {
                verify = false;
                rewalk = true;
                const bool on = ru >= u0 && ru < gcount;
                const int nbu = on ? (int) bt_n[rsc] : 0, nru = on ? (int) rc_n[rsc] : 0;
                if (nbu != 255 && rk < nbu)
                {
                    T.alive[bt_id[rsc][rk]] = 1;
                    T.comp[bt_id[rsc][rk]] = -1; // (not in the list of unfinished trees yet: see the exact finish prediction below)
                }
                wave_lds_fence();
                q_id = (nru != 255 && rk < nru) ? (int) rc_id[rsc][rk] : -1;
                q_fin = rc_fin[rsc][rk];
                const unsigned char al = T.alive[q_id >= 0 ? q_id : 0];
                const bool bad = on && (nru == 255 || nbu == 255 || (rk < nru && !al));
                q_id = al ? q_id : -1; // (a record of a dead tree makes its column a cut: nothing behind it is looked at)
                const unsigned long long bm = __ballot(bad);
                badmask = 0;
#pragma unroll
                for (int u = 0; u < G; u++)
                    if ((bm >> (u * A3_REC)) & ((1ull << A3_REC) - 1ull))
                        badmask |= 1u << u;
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
            // ---- which columns can let an EXISTING cluster through the finished-cluster check (cc.cpp:884-885: the cluster's largest
            // finished_at is not ahead of the column's smallest azimuth)? The scalar bound L (minimum over the clusters, as of the last
            // exact check) goes stale as soon as the cluster it came from receives points, and four of five checks it asked for retired
            // nothing. With at most 64 unfinished trees the prediction is made per cluster instead: the records of the group's columns
            // (one lane per (column, record)) raise their cluster's value column by column in a [column][list position] table (the
            // scratch of the finish pass), then one lane per tree walks its cluster's running maximum against the columns' azimuths.
            // Trees born inside the group are covered by the prefix minimum over the new roots' finished_at below, links only merge
            // clusters (never lower a maximum): a column that is not flagged cannot finish anything. -----------------------------
            unsigned long long m_alarm = 0;
            const bool exact = n_unf <= 64;
            if (exact && n_unf > 0)
            {
                static_assert(2 * TREE_SLOTS >= G * 64, "a_fin + a_min hold the [column][list position] table");
                unsigned long long* cf = &T.a_fin[0]; // [G][64], runs on into a_min (both are scratch of the finish pass)
#pragma unroll
                for (int u = 0; u < G; u++)
                    cf[u * 64 + lane] = 0ull;
                const int ti = T.alist[lane < n_unf ? lane : 0];
                if (lane < n_unf)
                    T.comp[ti] = lane; // list position of every unfinished tree
                wave_lds_fence();
                // (the loads of the record lanes and of the tree lanes are independent: one LDS round trip for both)
                const bool rec_on = q_id >= 0 && ru >= u0 && ru < gcount;
                const int qi = rec_on ? q_id : 0;
                const int pos0 = T.comp[qi];
                int rep = T.uf[qi];
                const int t_uf = T.uf[ti];
                unsigned long long run = T.c_fin[ti];
                if (rec_on && pos0 >= 0) // (an id born in this group has no list position yet: its tree is covered by the new-root bound)
                {
                    if (rep != qi)
                        rep = lds_find(T.uf, rep);
                    atomicMax(&cf[ru * 64 + T.comp[rep]], q_fin);
                }
                wave_lds_fence();
                const bool isrep = lane < n_unf && t_uf == ti;
                unsigned long long cfv[G];
#pragma unroll
                for (int u = 0; u < G; u++)
                    cfv[u] = cf[u * 64 + lane];
                unsigned hit = 0; // bit u: this lane's cluster is not ahead of column u's smallest azimuth
#pragma unroll
                for (int u = 0; u < G; u++)
                    if (u >= u0 && u < gcount)
                    {
                        run = cfv[u] > run ? cfv[u] : run;
                        const double mz = lane_f64(v_minaz, u);
                        hit |= !(__longlong_as_double((long long) run) > mz) ? (1u << u) : 0u;
                    }
                hit = isrep ? hit : 0u;
                if (__any(hit != 0)) // (most groups finish nothing)
                {
#pragma unroll
                    for (int u = 0; u < G; u++)
                        if (__ballot((hit >> u) & 1u))
                            m_alarm |= 1ull << u;
                }
            }
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
                const bool c_fin_hit = exact ? (((m_alarm >> wu) & 1ull) != 0 || v_minaz >= pm) : v_minaz >= w_L;
                const bool c_check = inr && w_nafter > 0 && !w_alias && ((gcu_l + 1 - w_M) >= NC || c_fin_hit);
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
                const bool c_fin_hit = exact ? (((m_alarm >> wu) & 1ull) != 0 || v_minaz >= pm) : v_minaz >= w_L;
                const bool c_check = inr && w_nafter > 0 && !w_alias && ((gcu_l + 1 - w_M) >= NC || c_fin_hit);
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
                const bool inb = ru >= u0 && ru < u1;
                const int nbu = inb ? (int) bt_n[rsc] : 0, nru = inb ? (int) rc_n[rsc] : 0; // (never 255 here: such a column is a cut)
                const int nb_before = __shfl(w_nbefore, ru);
                const long long gcr = gc + ru;
                if (rk < nbu)
                {
                    const int i = bt_id[rsc][rk];
                    int lcu = lc0 + ru;
                    lcu = lcu >= RC ? lcu - RC : lcu;
                    const unsigned long long f = bt_fin[rsc][rk];
                    T.cell[i] = lcu * R + (int) bt_row[rsc][rk];
                    T.gcol[i] = gcr;
                    T.fin[i] = f;
                    T.last[i] = gcr;
                    T.pts[i] = 1;
                    T.uf[i] = i;
                    T.c_fin[i] = f;
                    T.alist[nb_before + rk] = (short) i;
                    T.alive[i] = 1;
                }
                wave_lds_fence();
                if (rk < nru)
                {
                    const int i = rc_id[rsc][rk];
                    const unsigned long long f = rc_fin[rsc][rk];
                    const int rep = lds_find(T.uf, i);
                    atomicMax(&T.last[i], gcr);
                    atomicMax(&T.fin[i], f);
                    atomicMax(&T.c_fin[rep], f);
                    atomicAdd(&T.pts[i], (unsigned) rc_cnt[rsc][rk]);
                }
                wave_lds_fence();
                // links to other trees (rare: wave A flags the columns that have any): per point, from the staged inputs
                unsigned long long lm = __ballot(lane < G && lane >= u0 && lane < u1 && ((v_abad >> 4) & 1));
                while (lm)
                {
                    const int u = (int) __ffsll((long long) lm) - 1;
                    lm &= lm - 1;
                    const long long gcu = gc + u;
                    int lcu = lc0 + u;
                    lcu = lcu >= RC ? lcu - RC : lcu;
                    const int wcu = (int) (gcu & (WIN2_COLS - 1));
#pragma unroll
                    for (int k = 0; k < RPL; k++)
                    {
                        const int row = k * 64 + lane;
                        const int rr = row < R ? row : 0;
                        const int par = row < R ? (int) p.sc_parent[lcu * R + rr] : -2; // (rare path: straight from HBM)
                        const int e = row < R ? (int) s_win[wcu * R + rr] : -1;
                        const int nlk = (par >= 0 && row < R) ? (int) p.sc_nlinks[lcu * R + row] : 0; // (rare path: straight from HBM)
                        if (nlk > 0)
                        {
                            const int i = e & A2_IDMASK;
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

template<int RPL>
__global__ __launch_bounds__(A3_THREADS) void k_assoc3(Geometry g, cc_config cfg, Planes P, StreamState* states, int first_stream, int slot,
                                                       int limited, int count, int with_global)
{
    // with_global: streams whose unfinished trees do not fit the LDS pool (or that assoc3_stream just handed over) continue in global memory right here,
    // on wavefront 0 (associate_stream, cc_kernels.h), instead of in a launch of k_associate behind this kernel: one kernel boundary less on the
    // association chain (and one graph node less in a captured small call)
    if (gridDim.x >= (unsigned) count) // one block per stream
    {
        if ((int) blockIdx.x < count)
        {
            assoc3_stream<RPL>(g, cfg, P, states, first_stream + (int) blockIdx.x, slot, limited);
            if (with_global && !limited)
            {
                __threadfence_block();
                __syncthreads(); // every wavefront has left the stream (its state is in the planes again)
                if (threadIdx.x < 64)
                    associate_stream<RPL>(g, cfg, P, states, first_stream + (int) blockIdx.x, slot);
            }
        }
        return;
    }
    // a few blocks behind k_assocb: which streams have columns left is found out by all threads at once (one stream each: a block that walked over its
    // streams one after the other spent two dependent global round trips on each — 0.13 ms of the association chain per step at 64 streams, 20 % of the step),
    // then the block takes those streams one after the other — almost always none
    __shared__ int s_work[A3_THREADS];
    __shared__ int s_nwork;
    if (threadIdx.x == 0)
        s_nwork = 0;
    __syncthreads();
    for (int base = 0; base < count; base += (int) (gridDim.x * blockDim.x))
    {
        const int i = base + (int) threadIdx.x * (int) gridDim.x + (int) blockIdx.x;
        if (i < count)
        {
            const StreamState* st = &states[first_stream + i];
            if (st->error == 0 && st->batch[slot].seg_begin >= 0 && st->batch[slot].acp_next < st->batch[slot].seg_end)
                s_work[atomicAdd(&s_nwork, 1)] = i; // (at most blockDim.x entries per pass of `base`, and the list is drained before the next pass)
        }
        __syncthreads();
        const int nw = s_nwork;
        for (int k = 0; k < nw; k++)
        {
            assoc3_stream<RPL>(g, cfg, P, states, first_stream + s_work[k], slot, limited);
            __threadfence_block();
            __syncthreads(); // (the wavefronts leave a stream at different points; the LDS state is rebuilt from the planes for the next one)
            if (with_global && !limited)
            {
                if (threadIdx.x < 64)
                    associate_stream<RPL>(g, cfg, P, states, first_stream + s_work[k], slot);
                __syncthreads();
            }
        }
        if (threadIdx.x == 0)
            s_nwork = 0;
        __syncthreads();
    }
}
