// cc_k_publish.h — k_publish, k_small_tail, the frame scatter (k_scatter_info / k_scatter_apply), k_gather_clusters, k_view.
// (part of cc_kernels.h: included there, in order, inside namespace cck)
#pragma once

// =====================================================================================================
// k_publish — cluster ids of the columns published in this pass: Point::id = id of the finished cluster of the point's
// tree (cc.cpp:1005). grid = (PUBLISH_BLOCKS, streams), block = 64, lanes = rows.
// =====================================================================================================
constexpr int PUBLISH_BLOCKS = 64;

struct ViewOut
{
    float *x, *y, *z, *dist, *incl;
    double* caz;
    int64_t *gcol, *src, *root_gcol;
    uint8_t *ground, *debug, *ignored;
    uint64_t* id;
    int32_t* root_row;
    // the remaining clustering fields of Point (include/cc_hip.h), any of them may be null
    double* fin;
    uint32_t *tpts, *width, *nchild;
    int32_t *visits, *par_row;
    uint8_t* finished;
    int64_t* par_gcol;
};


// the host view of ONE column gc of a stream into column slot `oc` of the output arrays (k_view's body without the child counts; one wavefront)
__device__ __forceinline__ void view_column(const Geometry& g, const SP& p, const StreamState* st, const long long gc, const int oc, const ViewOut& o)
{
    const int R = g.num_rows, RC = g.ring_cols;
    const int lc = (int) (((gc % RC) + RC) % RC);
    const bool in_ring = st->ring_end >= 0 && gc >= 0 && gc >= st->clear_done && gc <= st->ring_end;
    const bool segmented = in_ring && st->first_column >= 0 && gc >= st->first_column && gc < st->first_unfinished;
    const CazBase cb = caz_base_of_column(gc >= 0 ? gc : 0, g.num_columns);
    const uint16_t tag = cell_tag((gc >= 0 ? gc : 0) / RC);
    for (int row = lane_id(); row < R; row += 64)
    {
        const size_t ci = (size_t) lc * R + row;
        const size_t oi = (size_t) oc * R + row;
        const float nanf_ = __builtin_nanf("");
        const bool mine = p.gtag[ci] == tag; // the cell belongs to this pass over the ring (Point::global_column_index == gc)
        const bool filled = in_ring && (segmented ? true : mine);
        const bool has_point = filled && !(p.dist[ci] != p.dist[ci]) && mine;
        const float4 rec = has_point ? p.sc_rec[ci] : make_float4(nanf_, nanf_, nanf_, nanf_);
        o.x[oi] = rec.x;
        o.y[oi] = rec.y;
        o.z[oi] = rec.z;
        o.dist[oi] = has_point ? p.dist[ci] : nanf_;
        o.incl[oi] = (has_point || segmented) ? p.incl[ci] : nanf_;
        // (a segmented cell without a return sits in the middle of its column, cc.cpp:371-372)
        o.caz[oi] = has_point ? cell_caz(cb, p.incaz[ci]) : (segmented ? empty_cell_caz(gc, g.az_width) : __builtin_nan(""));
        o.gcol[oi] = segmented ? gc : (has_point ? gc : -1);
        // (the firing's sequence number, kept as its low 32 bits: it is one of the last 2^32 firings the stream consumed)
        o.src[oi] = has_point ? (long long) (st->firings_consumed - (unsigned long long) (uint32_t) ((uint32_t) st->firings_consumed - p.src[ci])) : -1;
        o.ground[oi] = segmented ? p.ground[ci] : (uint8_t) CC_GP_UNKNOWN;
        o.debug[oi] = segmented ? p.debug[ci] : (uint8_t) CC_DBG_WHITE;
        o.ignored[oi] = segmented ? p.ignored[ci] : 0;
        const int r = segmented ? p.root[ci] : -1;
        o.id[oi] = r >= 0 ? (uint64_t) p.t_cid[r] : 0ull;
        o.root_gcol[oi] = r >= 0 ? p.colg[r / R] : -1;
        o.root_row[oi] = r >= 0 ? r % R : 0;
        // per-tree values live at the root cell (cc.cpp:666-671, 818-822, 933); everything else keeps its cleared value
        const bool is_root = r >= 0 && (size_t) r == ci;
        if (o.fin)
            o.fin[oi] = is_root ? p.t_fin[ci] : 0.;
        if (o.tpts)
            o.tpts[oi] = is_root ? p.t_pts[ci] : 0u;
        if (o.width)
            o.width[oi] = is_root ? p.t_width[ci] : 0u;
        if (o.finished)
            o.finished[oi] = is_root ? p.t_finished[ci] : (uint8_t) 0;
        if (o.visits)
            o.visits[oi] = (segmented && g.mirror_fields) ? (int32_t) p.sc_visits[ci] : 0;
        const int code = (segmented && r >= 0) ? (int) p.sc_parent[ci] : -1; // (columns back << 8) | row of the point whose child list holds this one
        if (o.par_gcol)
            o.par_gcol[oi] = code >= 0 ? gc - (code >> 8) : -1;
        if (o.par_row)
            o.par_row[oi] = code >= 0 ? (code & 0xff) : 0;
    }
}

// where the planes of a ViewOut lie in one staging block of n cells (8-byte planes first to keep alignment); `bytes` = what is used of it
__host__ __device__ inline ViewOut view_layout(char* base, const size_t n, size_t* bytes = nullptr)
{
    ViewOut o;
    o.caz = (double*) base;
    o.gcol = (int64_t*) (base + n * 8);
    o.src = (int64_t*) (base + n * 16);
    o.root_gcol = (int64_t*) (base + n * 24);
    o.id = (uint64_t*) (base + n * 32);
    o.fin = (double*) (base + n * 40);
    o.par_gcol = (int64_t*) (base + n * 48);
    char* b4 = base + n * 56;
    o.x = (float*) b4;
    o.y = (float*) (b4 + n * 4);
    o.z = (float*) (b4 + n * 8);
    o.dist = (float*) (b4 + n * 12);
    o.incl = (float*) (b4 + n * 16);
    o.root_row = (int32_t*) (b4 + n * 20);
    o.tpts = (uint32_t*) (b4 + n * 24);
    o.width = (uint32_t*) (b4 + n * 28);
    o.nchild = (uint32_t*) (b4 + n * 32);
    o.visits = (int32_t*) (b4 + n * 36);
    o.par_row = (int32_t*) (b4 + n * 40);
    char* b1 = b4 + n * 44;
    o.ground = (uint8_t*) b1;
    o.debug = (uint8_t*) (b1 + n);
    o.ignored = (uint8_t*) (b1 + 2 * n);
    o.finished = (uint8_t*) (b1 + 3 * n);
    if (bytes)
        *bytes = (size_t) ((char*) o.finished + n - base);
    return o;
}

constexpr int MV_COLS = 8; // columns a small call mirrors into pinned memory with its results (HostMirror::view)

// what a small call on the host path hands back (cc_engine.hip: add_firings_small), written straight into pinned host memory by the last kernel of
// the call instead of by three copy nodes of its graph: the stream's state, its first events, the early-stop counter
struct HostMirror
{
    StreamState* state;
    cc_event* events;
    int max_events;
    int* remaining;
    const int* d_remaining;
    unsigned long long* seq;   // pinned: the number of mirrored calls so far, written LAST (the host spins on it instead of synchronising the stream)
    unsigned long long* d_seq; // device: [0] that number, [1] blocks of the current launch that are through
    unsigned long long* tail_req; // pinned (k_small_all): the number of the call whose serial fall-backs the host has to launch (k_small_tail), else 0
    // round 6: the host views (cc_engine_read_columns' output) of the columns the call's events name — the columns it segmented and the columns it
    // published, if they are at most MV_COLS together — ride along, so that a front-end that keeps a mirror of range_image_ (the drop-in class: one
    // read per addFiring) needs no second kernel and no copy per call. view_hdr (pinned): [0] the call's number, [1] columns (-1: not mirrored),
    // [2 ..] their global indices; `view`: planes of MV_COLS * rows cells in pinned memory (view_layout). nullptr: not wanted.
    long long* view_hdr;
    char* view; // (one pointer: the planes' places follow from it — twenty-two pointers as kernel arguments cost the small-call kernels 140 scalar spills)
    int view_rows;
};

// cluster ids of the columns the batch published (cc.cpp:1035-1092: what publishing leaves in Point::id), columns by .. ny .. strided
__device__ __forceinline__ void publish_body(const Geometry& g, const Planes& P, const StreamState* states, const int s, const int slot, const int by,
                                             const int ny)
{
    const StreamState* st = &states[s];
    if (st->batch[slot].pub_begin < 0)
        return;
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols;
    int plc = (int) ((st->batch[slot].pub_begin + by) % RC);
    const int plc_step = (int) ((unsigned) ny % (unsigned) RC);
    for (long long pc = st->batch[slot].pub_begin + by; pc < st->batch[slot].pub_end;
         pc += ny, plc = (plc + plc_step >= RC ? plc + plc_step - RC : plc + plc_step))
    {
        for (int row = lane_id(); row < R; row += 64)
        {
            const int ci = plc * R + row;
            const int r = p.root[ci];
            p.id[ci] = r >= 0 ? p.t_cid[r] : 0u;
        }
    }
}

// one wavefront: the call's results into pinned host memory (mirror_copy), then — behind a system-scope fence of every wavefront that wrote
// something the host will read — the sequence number (mirror_commit). Returns the number the call gets.
__device__ __forceinline__ unsigned long long mirror_copy(const Geometry& g, const Planes& P, const StreamState* states, const int s, const HostMirror& hm,
                                                          const int n_view, const long long seg_b, const int n_seg, const long long pub_b)
{
    const int lane = lane_id();
    const StreamState* s0 = &states[s];
    unsigned long long seq0 = 0ull; // (requested ahead of the copies: the last store of the call waits for nothing but the fence)
    if (lane == 0)
        seq0 = hm.d_seq[0];
    seq0 = (unsigned long long) uniform_i64((long long) seq0);
    const unsigned* src = (const unsigned*) s0;
    unsigned* dst = (unsigned*) hm.state;
    for (int i = lane; i < (int) (sizeof(StreamState) / 4); i += 64)
        dst[i] = src[i];
    const int ne = s0->n_events < hm.max_events ? s0->n_events : hm.max_events;
    const unsigned* es = (const unsigned*) (P.events + (size_t) s * g.event_capacity);
    unsigned* ed = (unsigned*) hm.events;
    for (int i = lane; i < ne * (int) (sizeof(cc_event) / 4); i += 64)
        ed[i] = es[i];
    if (lane == 0)
        *hm.remaining = *hm.d_remaining;
    if (hm.view_hdr)
    {
        if (lane == 0)
            hm.view_hdr[1] = n_view; // ([0], the call's number, is stamped by the last view writer once the views are out)
        if (lane < n_view)
            hm.view_hdr[2 + lane] = lane < n_seg ? seg_b + lane : pub_b + (lane - n_seg);
    }
    return seq0;
}

__device__ __forceinline__ void mirror_commit(const HostMirror& hm, const unsigned long long seq0)
{
    __threadfence_system();
    if (lane_id() == 0)
    {
        hm.d_seq[1] = 0ull;
        const unsigned long long v = seq0 + 1ull;
        hm.d_seq[0] = v;
        __hip_atomic_store(hm.seq, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__device__ __forceinline__ void mirror_results(const Geometry& g, const Planes& P, const StreamState* states, const int s, const HostMirror& hm)
{
    mirror_commit(hm, mirror_copy(g, P, states, s, hm, -1, 0, 0, 0));
}

__global__ __launch_bounds__(64) void k_publish(Geometry g, Planes P, const StreamState* states, int first_stream, int slot, HostMirror hm)
{
    publish_body(g, P, states, first_stream + (int) blockIdx.x, slot, (int) blockIdx.y, (int) gridDim.y);
    if (hm.state)
    {
        // the LAST block of the launch to get here mirrors the call's results: every cluster id of the call has been written by then, and the
        // association chain in front of this kernel left the state final
        __threadfence();
        unsigned long long through = 0;
        if (lane_id() == 0)
            through = atomicAdd(&hm.d_seq[1], 1ull);
        through = (unsigned long long) uniform_i64((long long) through);
        if (through == (unsigned long long) gridDim.x * gridDim.y - 1ull)
            mirror_results(g, P, states, first_stream, hm);
    }
}

// k_small_tail — what is behind the batch-parallel association in a call of a few firings on ONE stream (the per-column latency path): the exact serial
// kernel for whatever k_assocb left (nothing, normally), the streams that continue in global memory, the cluster ids of the published columns and
// the results into pinned host memory — k_assoc3 + k_publish in one launch (one graph node less: ~4.5 us of a 50 us call). grid = 1, block = A3_THREADS.
template<int RPL>
__global__ __launch_bounds__(A3_THREADS) void k_small_tail(Geometry g, cc_config cfg, Planes P, StreamState* states, int stream, int slot, HostMirror hm)
{
    assoc3_stream<RPL>(g, cfg, P, states, stream, slot, 0);
    __threadfence_block();
    __syncthreads(); // every wavefront has left the stream (its state is in the planes again)
    if (threadIdx.x < 64)
        associate_stream<RPL>(g, cfg, P, states, stream, slot);
    __syncthreads();
    publish_body(g, P, states, stream, slot, uniform_i32((int) (threadIdx.x >> 6)), (int) (blockDim.x >> 6));
    __syncthreads();
    if (hm.state && threadIdx.x < 64)
        mirror_results(g, P, states, stream, hm);
}

// =====================================================================================================
// k_small_all — a call of a few firings on ONE stream of a 64-row engine (the per-column latency path, cc_engine_add_firings with n < 64) in ONE
// launch: what k_small_front, k_assocb and k_small_tail do in three. A kernel node of a captured graph costs ~4.5 us of dispatch on a call whose
// kernels need 5 - 18 us each. grid = 1, block = AB_THREADS (k_assocb's block: 15 worker wavefronts + the timeline), dynamic LDS =
// insert2_lds_bytes(num_rows).
//   A - D  k_small_front's phases (begin, ego records, preparation: all threads; serial insertion: wavefronts 0 - 1; segmentation: wavefront 0;
//          window scan: one wavefront per column) — everything they hand over goes through global memory, block barriers in between
//   E      assocb_body: the batch-parallel association + finished-cluster check of the call's columns
//   F      nothing left for the serial kernels (the normal case): cluster ids of the published columns, results into pinned host memory, the
//          sequence number last. Otherwise (a stop of the batch-parallel kernel, a stream that continues in global memory) the call's number goes
//          into HostMirror::tail_req and the host launches k_small_tail behind this kernel, which then does all of F.
// =====================================================================================================
// the whole call (phases A - F above) as a device function: k_small_all is one invocation of it, k_resident a loop over it. Returns (to every
// thread alike) 0 when the call's results have been mirrored into pinned host memory, 1 when the serial fall-backs are needed: HostMirror::tail_req
// then names the call and whoever launched this must launch k_small_tail behind it.
__device__ __forceinline__ int small_all_body(const Geometry& g, const cc_config& cfg, const Planes& P, StreamState* states, const int stream, const int slot,
                                              const float* xyz, const uint8_t* inten, const double* poses, const long long n, int* remaining, double* ego,
                                              int* bail_count, const HostMirror& hm, AbShared<1>& S)
{
    const int R = g.num_rows;
    StreamState* st = &states[stream];
    const int wave = uniform_i32((int) (threadIdx.x >> 6));
#ifdef CC_SF_STATS
    unsigned long long sa_t[8];
    sa_t[0] = __builtin_amdgcn_s_memtime();
#define SA_MARK(i) sa_t[i] = __builtin_amdgcn_s_memtime();
#else
#define SA_MARK(i)
#endif
    // what the association will start from, as far as it can be known before the insertion has run: the call's first finished column is the
    // stream's first unfinished one (insert2_body closes the batch descriptor with it); assocb_body checks the prediction
    if (threadIdx.x == 0)
        S.s_view_done = 0; // (phase F: view writers that are through; several block barriers lie between here and there)
    const unsigned long long call_no = hm.d_seq ? (unsigned long long) uniform_i64((long long) hm.d_seq[0]) + 1ull : 0ull; // (read before wavefront 0 moves it on)
    AbPreloaded pre;
    pre.col_begin = st->first_unfinished;
    pre.first_column = st->first_column;
    pre.n_unf = st->n_unfinished;
    pre.valid = (pre.n_unf >= 0 && pre.n_unf <= AB_TREES) ? 1 : 0;
    if (threadIdx.x >= 256 && pre.valid)
        assocb_load_state<1>(g, stream_ptrs(P, g, stream), S, pre.col_begin, pre.first_column, pre.n_unf, (int) threadIdx.x - 256, AB_THREADS - 256);
    // (three wavefronts side by side: the batch begins — wavefront 2 —, the firings' ego records — wavefront 1, one lane per firing: two 3 x 4
    // products in f64 behind a read of pinned host memory —, the points — wavefronts 0 - 3, wavefront 0 first. In one wavefront they ran one
    // after the other: three round trips to host memory instead of one)
    if (threadIdx.x == 128)
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
    for (long long f = (long long) threadIdx.x - 64; f >= 0 && f < n && threadIdx.x < 128; f += 64)
        ego_record(states, stream, cfg, poses, n, n, 0, ego, 0, f);
    for (long long i = threadIdx.x; i < n * R && threadIdx.x < 256; i += 256)
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
        long long* w = insert2_sync_words<1>(R);
        lds_st(w, 0ll);
        lds_st(w + 1, 0ll);
        lds_st(w + 2, -1ll);
    }
    __syncthreads();
    SA_MARK(1)
    if (threadIdx.x < 128)
        insert2_body<1, true, true>(g, cfg, P, states, stream, slot, inten, n, remaining, n, 0, 0);
    else if (wave == 4 && pre.valid)
        assocb_representatives<1>(S, pre.n_unf);
    __syncthreads();
    SA_MARK(2)
    if (threadIdx.x < 64)
        seg_small_body(g, cfg, P, states, stream, slot, poses, n, 0, ego, n, 0);
    __syncthreads();
    SA_MARK(3)
    if (g.mirror_fields)
        scan_body<1, true>(g, cfg, P, states, stream, slot, 0, wave, AB_WAVES + 1);
    else
        scan_body<1, false>(g, cfg, P, states, stream, slot, 0, wave, AB_WAVES + 1);
    __threadfence_block();
    __syncthreads();
    SA_MARK(4)
    assocb_body<1>(g, cfg, P, states, stream, slot, bail_count, S, pre);
    __threadfence_block();
    __syncthreads();
    SA_MARK(5)
    // F: is anything left for the serial kernels (assoc3_stream / associate_stream take a stream under exactly these conditions)?
    const bool left = uniform_i32((int) (st->error == 0 && st->batch[slot].seg_begin >= 0 && st->batch[slot].acp_next < st->batch[slot].seg_end)) != 0;
    if (left)
    {
        if (threadIdx.x == 0)
            __hip_atomic_store(hm.tail_req, hm.d_seq[0] + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return 1;
    }
    // (the mirror carries the stream's state and events, final since assocb_body: wavefront 0 sends it while the others write the cluster ids,
    // which stay in HBM and are complete when the kernel ends — whatever reads them is ordered behind it on the stream — and the host views of the
    // columns the call's events name: what it segmented, what it published)
    const long long seg_b = st->batch[slot].seg_begin, seg_e = st->batch[slot].seg_end, pub_b = st->batch[slot].pub_begin, pub_e = st->batch[slot].pub_end;
    const int n_seg = uniform_i32(seg_b >= 0 && seg_e > seg_b ? (int) (seg_e - seg_b < 64 ? seg_e - seg_b : 64) : 0);
    const int n_pub = uniform_i32(pub_b >= 0 && pub_e > pub_b ? (int) (pub_e - pub_b < 64 ? pub_e - pub_b : 64) : 0);
    const int n_view = (hm.view_hdr && n_seg + n_pub <= MV_COLS && st->error == 0) ? n_seg + n_pub : -1;
    // Wavefront 0 names the call as soon as ITS copies are out (state, events, and the header that says which views are coming); the view writers
    // finish a few microseconds later and the LAST of them stamps the header with the call's number: cc_engine_read_columns waits for that stamp
    // — by then the host has usually not even finished reading the events. (With the views in front of the call's number every call paid for
    // them: + 5 us; with a block barrier as well for the wavefronts that write cluster ids: + 10 us.)
    const int n_writers = n_view > 0 ? (n_view < 4 ? n_view : 4) : 0;
    if (wave == 0)
        mirror_commit(hm, mirror_copy(g, P, states, stream, hm, n_view, seg_b, n_seg, pub_b));
    else
    {
        if (wave <= n_writers)
        {
            const SP pv = stream_ptrs(P, g, stream);
            ViewOut vo = view_layout(hm.view, (size_t) MV_COLS * (size_t) hm.view_rows);
            vo.nchild = nullptr;
            for (int j = wave - 1; j < n_view; j += n_writers)
                view_column(g, pv, st, j < n_seg ? seg_b + j : pub_b + (j - n_seg), j, vo);
            __threadfence_system(); // (this wavefront's stores to pinned memory are out before it says so)
            int last = 0;
            if (lane_id() == 0)
                last = __hip_atomic_fetch_add(&S.s_view_done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == n_writers - 1 ? 1 : 0;
            if (last)
                __hip_atomic_store(&hm.view_hdr[0], (long long) call_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        publish_body(g, P, states, stream, slot, wave - 1, AB_WAVES);
    }
#ifdef CC_SF_STATS
    SA_MARK(6)
    if (threadIdx.x == 0)
    {
        // (read by tools/sf_probe.py; the mirror above has gone out, so these reach the host with the NEXT call's state)
        for (int i = 0; i < 6; i++)
            st->dbg[i] += sa_t[i + 1] - sa_t[i];
        st->dbg[6] += 1;
    }
#endif
    return 0;
}

__global__ __launch_bounds__(AB_THREADS) void k_small_all(Geometry g, cc_config cfg, Planes P, StreamState* states, int stream, int slot,
                                                          const float* __restrict__ xyz, const uint8_t* __restrict__ inten, const double* __restrict__ poses,
                                                          long long n, int* remaining, double* __restrict__ ego, int* __restrict__ bail_count, HostMirror hm)
{
    __shared__ AbShared<1> S;
    (void) small_all_body(g, cfg, P, states, stream, slot, xyz, inten, poses, n, remaining, ego, bail_count, hm, S);
}

// =====================================================================================================
// k_resident — the synchronous calling pattern of the reference (is_single_threaded: addFiring returns when the firing's columns are through,
// thread_pool.hpp:58-64, cc.cpp:88-93; kitti_demo.cpp:280,403) without a kernel dispatch per call: ONE block of AB_THREADS threads stays on a
// compute unit and runs small_all_body once per doorbell. Host and kernel talk through pinned host memory (ResidentCtl):
//   host    copies the call's firings into the pinned staging (xyz / inten / poses, the same buffers k_small_all reads), then stores
//           bell = call number << 8 | n  (release, system scope) and spins on HostMirror::seq / tail_req / ResidentCtl::exited
//   kernel  thread 0 polls `bell` (one load over PCIe per ~1 us) until it names the next call; the block runs the call; results go to the pinned
//           mirror as in k_small_all, the sequence number last. Writes to the planes are released to the device (the system-scope fence in
//           mirror_results writes the L2 back), so read-only queries on ANOTHER stream (k_view, k_gather_clusters) see them while this kernel idles.
// The kernel leaves (ResidentCtl::exited = reason, system scope, last store) when
//   1  the host asked for it (ResidentCtl::stop: reset, set_config, set_option, a call of another shape, destroy ...)
//   2  a call needs the serial fall-backs (tail_req names it: the host launches k_small_tail behind this kernel, in stream order)
//   3  a call stopped early (limit_columns: the host runs the continuation passes) or left an error on the stream
//   4  WATCHDOG: no doorbell for `idle_limit` ticks of the 100 MHz wall clock (a host thread that died or went away must not pin a compute
//      unit and a PCIe poll loop forever); the next call simply launches the kernel again
// and the host waits for the kernel's end on the stream before it launches anything else there. A doorbell that arrives while the kernel is
// leaving is not lost: `bell` keeps its value and the next launch of the kernel takes it.
// grid = 1, block = AB_THREADS, dynamic LDS = insert2_lds_bytes(num_rows).
// =====================================================================================================
struct ResidentCtl
{
    unsigned long long bell;   // host -> kernel: call number << 8 | firings (1 .. 63)
    unsigned long long stop;   // host -> kernel: leave now (next to the bell: the kernel reads both with ONE 16-byte load over PCIe)
    unsigned long long pad0[6];
    unsigned long long pad1[8];
    unsigned long long exited; // kernel -> host: 0 while it runs, else the reason it left
    unsigned long long calls;  // kernel -> host: calls this launch of the kernel has run (statistics)
    unsigned long long pad2[6];
};

__global__ __launch_bounds__(AB_THREADS) void k_resident(Geometry g, cc_config cfg, Planes P, StreamState* states, int stream, int slot, const float* xyz,
                                                         const uint8_t* inten, const double* poses, int* remaining, double* ego, int* bail_count, HostMirror hm,
                                                         ResidentCtl* ctl, unsigned long long idle_limit)
{
    __shared__ AbShared<1> S;
    __shared__ long long s_cmd; // > 0: firings of the next call; < 0: leave with reason -s_cmd
    unsigned long long calls = 0ull;
    for (;;)
    {
        if (threadIdx.x == 0)
        {
            const unsigned long long want = hm.d_seq[0] + 1ull; // (written by this block's own mirror_results, or by the launch before)
            const unsigned long long t0 = wall_clock64();
            long long cmd = 0;
            // A poll is a read of host memory over PCIe (~1.5 us there and back). Four are kept in flight, a quarter of a microsecond apart: the
            // bell is seen one trip after the host's store instead of one and a half to two (one read at a time, bell and stop flag one after the other)
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4* bp = (const u32x4*) &ctl->bell;
            for (unsigned spin = 0; cmd == 0; spin++)
            {
                u32x4 v[4];
                // (volatile or atomic loads in C are waited for one by one; four system-coherent loads in flight need the instruction sequence spelled out)
                asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\ts_sleep 4\n\t"
                             "global_load_dwordx4 %1, %4, off sc0 sc1\n\ts_sleep 4\n\t"
                             "global_load_dwordx4 %2, %4, off sc0 sc1\n\ts_sleep 4\n\t"
                             "global_load_dwordx4 %3, %4, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                             : "v"(bp)
                             : "memory");
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const unsigned long long b = (unsigned long long) v[k].x | ((unsigned long long) v[k].y << 32);
                    const unsigned long long stopv = (unsigned long long) v[k].z | ((unsigned long long) v[k].w << 32);
                    if (cmd == 0 && (b >> 8) == want && (b & 255ull) != 0ull)
                        cmd = (long long) (b & 255ull);
                    if (cmd == 0 && stopv != 0ull)
                        cmd = -1;
                }
                if (cmd == 0 && (spin & 7u) == 7u && wall_clock64() - t0 > idle_limit)
                    cmd = -4;
            }
            s_cmd = cmd;
        }
        __syncthreads();
        const long long cmd = uniform_i64(s_cmd);
        int reason = cmd < 0 ? (int) -cmd : 0;
        if (reason == 0)
        {
            // acquire at system scope, every wavefront: what the host wrote before it rang the bell (the firings in pinned memory) is read from
            // memory, not from a cache line of the previous call, and no load of the call is moved in front of the poll
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            __builtin_amdgcn_s_dcache_inv();
            const int left = small_all_body(g, cfg, P, states, stream, slot, xyz, inten, poses, cmd, remaining, ego, bail_count, hm, S);
            calls++;
            __syncthreads(); // (S and the insertion's LDS words are re-used by the next call)
            if (left)
                reason = 2;
            else if (uniform_i32((int) (*remaining != 0 || states[stream].error != 0)) != 0)
                reason = 3;
        }
        if (reason)
        {
            if (threadIdx.x == 0)
            {
                __hip_atomic_store(&ctl->calls, calls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __threadfence_system();
                __hip_atomic_store(&ctl->exited, (unsigned long long) reason, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
    }
}

// =====================================================================================================
// k_scatter_info / k_scatter_apply — the frame scatter of the reference's harness (addColumnAndEvaluateFrameIfCompleted,
// kitti_demo.cpp:173-224) for a replayed KITTI sequence, on the device. A stream that is fed exactly num_columns pseudo-firings per frame
// (kitti_demo.cpp:386-403) carries, per cell, the sequence number of the firing that filled it: frame = sequence / num_columns, range-image
// column of the frame = sequence % num_columns, and the KITTI point of the cell is original_index[frame % slots][column][row]
// (cc_kitti_frame::d_original_index of the frame's conversion). k_scatter_info gives the smallest / largest frame among the points of every
// published column (what the harness needs to find where frame N + 1 starts, :205-209, and its two error conditions); k_scatter_apply
// writes is_ground_point = (ground_point_label == GP_GROUND) and detection_label = id (:214-215) of the columns' points into the frames'
// arrays in HBM, which cc_eval_frame_device then reads. grid = columns, block = 64 (lanes = rows).
// =====================================================================================================
__global__ __launch_bounds__(64) void k_scatter_info(Geometry g, Planes P, const StreamState* __restrict__ states, int s, long long from,
                                                     const int* __restrict__ original_index, int slots, int* __restrict__ out_min,
                                                     int* __restrict__ out_max)
{
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols, NC = g.num_columns;
    const long long gc = from + blockIdx.x;
    const int lc = (int) (gc % RC);
    const int* org = original_index + (size_t) s * (size_t) slots * (size_t) NC * (size_t) R;
    int mn = 0x7fffffff, mx = -1;
    // only published columns that are still in the ring hold what this reads (anything else: "no point", like an empty column)
    // (clearing is deferred by one call, so what a call published stays readable behind ring_start: the lower end is what has been CLEARED)
    const bool live = gc >= 0 && gc >= states[s].clear_done && gc < states[s].first_unpublished;
    for (int row = lane_id(); live && row < R; row += 64)
    {
        const int ci = lc * R + row;
        if (p.dist[ci] == p.dist[ci]) // the cell holds a return
        {
            const unsigned seq = p.src[ci];
            const int frame = (int) (seq / (unsigned) NC), col = (int) (seq % (unsigned) NC);
            if (org[((size_t) (frame % slots) * NC + col) * R + row] >= 0)
            {
                mn = frame < mn ? frame : mn;
                mx = frame > mx ? frame : mx;
            }
        }
    }
    mn = wave_min_i32(mn);
    mx = -wave_min_i32(-mx);
    if (lane_id() == 0)
    {
        out_min[blockIdx.x] = mn;
        out_max[blockIdx.x] = mx;
    }
}

__global__ __launch_bounds__(64) void k_scatter_apply(Geometry g, Planes P, const StreamState* __restrict__ states, int s, long long from,
                                                      const int* __restrict__ original_index, int slots, unsigned char* __restrict__ is_ground,
                                                      unsigned* __restrict__ detection, long long max_points)
{
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols, NC = g.num_columns;
    const long long gc = from + blockIdx.x;
    if (gc < 0 || gc < states[s].clear_done || gc >= states[s].first_unpublished)
        return; // (not a published column of the live ring)
    const int lc = (int) (gc % RC);
    const int* org = original_index + (size_t) s * (size_t) slots * (size_t) NC * (size_t) R;
    unsigned char* gr = is_ground + (size_t) s * (size_t) slots * (size_t) max_points;
    unsigned* det = detection + (size_t) s * (size_t) slots * (size_t) max_points;
    for (int row = lane_id(); row < R; row += 64)
    {
        const int ci = lc * R + row;
        if (p.dist[ci] == p.dist[ci])
        {
            const unsigned seq = p.src[ci];
            const int frame = (int) (seq / (unsigned) NC), col = (int) (seq % (unsigned) NC);
            const int pt = org[((size_t) (frame % slots) * NC + col) * R + row];
            if (pt >= 0 && pt < max_points)
            {
                const size_t o = (size_t) (frame % slots) * (size_t) max_points + (size_t) pt;
                gr[o] = p.ground[ci] == CC_GP_GROUND ? 1 : 0;
                det[o] = p.id[ci];
            }
        }
    }
}

// =====================================================================================================
// k_gather_clusters — member points of finished clusters, compacted on the device (the point gathering of
// collectPointsForCusterAndPublish, cc.cpp:985-1033): cluster i owns out[offset[i] .. offset[i] + n_points[i]) and receives its
// points in (global column, row) order. grid = clusters, block = 64 (lanes = rows), one pass over the cluster's column range.
// A point belongs to cluster c iff the root of its point tree carries c (t_cid, set when the cluster is finished).
// =====================================================================================================
struct ClusterQuery
{
    const unsigned* cid;       // [n] cluster ids (CC_EV_CLUSTER.c)
    const long long* col_from; // [n] first column (CC_EV_CLUSTER.a)
    const long long* col_to;   // [n] last column (CC_EV_CLUSTER.b)
    const long long* offset;   // [n] first output element of the cluster
    const unsigned* n_points;  // [n] expected number of points (CC_EV_CLUSTER.d)
    long long* out_gcol;
    int* out_row;
    int* mismatch; // incremented per cluster whose point count differs from n_points (columns cleared already, wrong descriptor)
};

__global__ __launch_bounds__(64) void k_gather_clusters(Geometry g, Planes P, const StreamState* states, int s, ClusterQuery q)
{
    const int ci_ = blockIdx.x;
    const SP p = stream_ptrs(P, g, s);
    const StreamState* st = &states[s];
    const int R = g.num_rows, RC = g.ring_cols;
    const int lane = lane_id();
    const unsigned cid = q.cid[ci_];
    const long long a = q.col_from[ci_], b = q.col_to[ci_];
    long long pos = q.offset[ci_];
    const long long end = pos + q.n_points[ci_];
    const bool readable = cid != 0 && a >= 0 && b >= a && b - a < RC && a >= st->clear_done && b <= st->ring_end;
    if (readable)
    {
        int lc = (int) (a % RC);
        for (long long gc = a; gc <= b; gc++, lc = (lc + 1 == RC ? 0 : lc + 1))
            for (int r0 = 0; r0 < R; r0 += 64)
            {
                const int row = r0 + lane;
                bool mine = false;
                if (row < R)
                {
                    const int cell = lc * R + row;
                    const int root = p.root[cell];
                    mine = p.colg[lc] == gc && root >= 0 && p.t_cid[root] == cid && p.t_finished[root];
                }
                const unsigned long long mask = __ballot(mine);
                if (mine)
                {
                    const long long o = pos + __popcll(mask & lanes_below());
                    if (o < end)
                    {
                        q.out_gcol[o] = gc;
                        q.out_row[o] = row;
                    }
                }
                pos += __popcll(mask);
            }
    }
    if (lane == 0 && pos != end)
        atomicAdd(q.mismatch, 1);
}

// =====================================================================================================
// k_view — host view of columns [from, from + ncols) of one stream (cc_engine_read_columns)
// grid = ncols, block = 64
// =====================================================================================================
// (up to 8 ranges of columns per launch: block b views column from[r] + (b - start[r]) of the range r it falls into; the output holds the ranges one after the other)
struct ViewRanges
{
    int n;
    int start[9];
    long long from[8];
};

__global__ __launch_bounds__(64) void k_view(Geometry g, Planes P, const StreamState* states, int s, ViewRanges vr, ViewOut o, int max_back)
{
    const StreamState* st = &states[s];
    const SP p = stream_ptrs(P, g, s);
    const int R = g.num_rows, RC = g.ring_cols;
    int r = 0;
    while (r + 1 < vr.n && (int) blockIdx.x >= vr.start[r + 1])
        r++;
    const long long gc = vr.from[r] + ((int) blockIdx.x - vr.start[r]);
    const int lc = (int) (((gc % RC) + RC) % RC);
    const bool in_ring = st->ring_end >= 0 && gc >= 0 && gc >= st->clear_done && gc <= st->ring_end;
    const bool segmented = in_ring && st->first_column >= 0 && gc >= st->first_column && gc < st->first_unfinished;
    view_column(g, p, st, gc, (int) blockIdx.x, o);
    if (o.nchild)
    {
        // Point::child_points.size(): the points of this and the following columns whose parent is a cell of this column
        __shared__ unsigned s_cnt[WAVE * MAX_ROWS_PER_LANE];
        for (int row = lane_id(); row < R; row += 64)
            s_cnt[row] = 0;
        __syncthreads();
        if (segmented)
            for (int d = 0; d <= max_back; d++)
            {
                const long long gd = gc + d;
                if (gd >= st->first_unfinished)
                    break;
                int ld = lc + d;
                ld = ld >= RC ? ld - RC : ld;
                for (int row = lane_id(); row < R; row += 64)
                {
                    const size_t cj = (size_t) ld * R + row;
                    const int code = p.root[cj] >= 0 ? (int) p.sc_parent[cj] : -1;
                    if (code >= 0 && (code >> 8) == d)
                        atomicAdd(&s_cnt[code & 0xff], 1u);
                }
            }
        __syncthreads();
        for (int row = lane_id(); row < R; row += 64)
            o.nchild[(size_t) blockIdx.x * R + row] = s_cnt[row];
    }
}
