// cc_engine.hip — host side of the C-ABI in include/cc_hip.h: HBM allocation, batched launches, D2H views.
// Compiled by hipcc for gfx950 only; there is no CPU implementation behind this ABI — without a GPU
// cc_engine_create fails with CC_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "cc_kernels.h"

// launches of at most this many streams run k_assoc3 with its links wavefront when the number of association wavefronts is automatic
#ifndef CC_LWAVE_MAX_STREAMS
#define CC_LWAVE_MAX_STREAMS 256
#endif

using namespace ccd;

struct cc_engine
{
    int device{0};
    Geometry g{};
    cc_config cfg{};
    Planes P{};
    StreamState* d_states{nullptr};
    int* d_remaining{nullptr};
    int* h_remaining{nullptr}; // pinned
    int* d_bail_count{nullptr}; // launches of k_assocb that stopped in front of a group, ever (assoc_rounds 0 = adaptive)
    int* h_bail_count{nullptr}; // pinned; refreshed behind every batch's association chain
    int bail_seen{0}, bail_cooldown{0};
    int chronic_seen{0}, chronic_skip{0}; // batches left in which the serial association kernels run alone (k_assocb kept stopping)
    bool chronic_probe{false};
    int assoc_sweep_blocks{2};     // option "assoc_sweep_blocks": blocks of the serial kernel's launch while it is only the safety net behind k_assocb (16 at first: the fewer
                                   // 256-thread / 50 KB blocks have to be placed next to the other chains, the sooner the next batch's k_assocb starts: 0.2 -> 0.1 ms at 256 streams)
    int bail_cooldown_batches{4};  // option "assoc_cooldown": batches that run three (batch-parallel, serial) rounds after k_assocb had to stop (assoc_rounds 0).
                                   // 16 in the first version: with the few stops of ordinary streams (5 in 47 M columns) nearly every batch of a 256-stream run
                                   // then ran three rounds — two more placements of k_assocb's 1024-thread blocks per step, 2 - 3 % of the step
    bool capturing{false};      // launch_batch is being captured into a hipGraph (small calls)
    cck::HostMirror capture_mirror{nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr}; // set while such a graph is captured with the results mirrored by k_publish
    int* d_par_left{nullptr};  // [0] streams whose batch k_insert_par did not take completely (skip_idle_fallbacks), [1] streams whose batch still needs
                               // k_table / k_seg_pre (not closed as fused)
    bool small_front{true};    // option "small_front": a call of < 64 firings on one stream of a 64-row engine runs k_small_front (begin + ego + prep + insertion + segmentation in one launch)
    bool small_all{true};      // option "small_all": such a call whose results are mirrored into pinned memory is ONE launch (k_small_all) instead of k_small_front + k_assocb + k_small_tail
    bool small_direct{true};   // option "small_direct": calls of 9 .. 63 firings on one stream are ONE direct launch of k_small_all (up to 8: a captured one-node graph; 0: the general path)
    int seg_small_max{63};     // option "seg_small_max": calls of at most this many firings (64-row sensors) segment their columns with k_seg_small
    bool fuse_front{true};     // option "fuse_front": k_insert_par also does the per-cell part of the segmentation of the columns it fills
    int* h_par_left{nullptr};  // pinned
    int insert_split_blocks{0};      // option "insert_split_blocks": blocks per stream of k_insert_par in such launches (0 = 8 up to 24 streams, 6 up to 32, 4 up to 40, 3 up to 64, else 2; 1 = one)
    int insert_narrow_blocks{0};     // option "insert_narrow_blocks": above insert_wide_max_streams, blocks of 4 wavefronts, this many per stream (0 = one block of 8)
    int insert_wide_max_streams{160}; // option "insert_wide_max_streams": launches of at most this many streams run k_insert_par with 16 wavefronts
    bool skip_idle_fallbacks{true}; // option "skip_idle_fallbacks": wait for k_insert_par and launch the other insertion kernels only if needed
    hipStream_t stream{nullptr};  // insertion chain (and everything else when not pipelined)
    hipStream_t stream2{nullptr}; // table / segmentation / window-scan chain of the pipelined throughput path
    hipStream_t stream3{nullptr}; // association / publish chain of the pipelined throughput path
    hipStream_t stream4{nullptr}; // window-scan stage of the four-stage pipeline (option "pipeline" = 2)
    hipStream_t stream7{nullptr}; // k_table with table_on_insert_chain = 2
    hipStream_t stream6{nullptr}; // k_publish of a pipelined batch: off the association chain, which is the longest of the three
    hipEvent_t ev_pubrdy[4]{};
    hipEvent_t ev_ego[4]{};       // k_ego of the slot's batch on the preparation stream (option "ego_off_chain")
    bool ego_off_chain{false};    // (measured, round 6: 32 streams - 8 % in the 20-step leg and - 12 % steady with it on — the cross-stream event costs more than the kernel's ~10 us on the chain —, 256 streams + 0)
    hipStream_t stream5{nullptr}; // k_prep of the *next* batch: independent of the engine state, so it runs ahead of the insertion chain
    hipEvent_t ev_ins[4]{}, ev_seg[4]{}, ev_assoc[4]{}, ev_segscan[4]{}, ev_prep[4]{};
    hipEvent_t ev_input{};            // option "input_on_engine_stream": recorded on `stream` when a device call arrives
    int pipeline_depth{2};        // 2 (default since round 3: with the batch-parallel association the segmentation + scan chain became the longest, +10 %
                                  // at 256 streams with the window scan on a chain of its own): four chains, 1: three
    int prep_buf{0};              // staging buffer (of two) the open batch was prepared into
    std::vector<hipEvent_t> pev_pool; // pairs of events around every k_prep that ran ahead (outside the per-pass event groups)
    size_t pev_used{0};
    struct PrepPlanes
    {
        float *x, *y, *z, *dist, *incl, *incaz;
        int32_t* cir;
    } pp[2]{};
    uint64_t batch_seq{0};        // batches submitted since reset; slot = batch_seq & 3
    // ---- lifetime of the caller's device buffers (cc_engine_inputs_released, include/cc_hip.h) --------------------------------------------
    // A call of cc_engine_add_firings_device has a number (1, 2, ...). Its inputs are read for the last time by kernels that are all ordered in
    // front of the end of the batch's publishing chain, so an event recorded behind that chain (ev_rel, a ring of REL_RING) says "released";
    // a full synchronisation at the API level releases everything submitted. Nothing here waits.
    static constexpr int REL_RING = 8;
    uint64_t call_seq{0};               // calls of cc_engine_add_firings_device so far (never reset: the numbers stay unique over cc_engine_reset)
    uint64_t released_seq{0};           // ... whose inputs will not be read again, as far as the host has noticed
    hipEvent_t ev_rel[REL_RING]{};
    uint64_t rel_recorded[REL_RING]{};  // the call whose end the event was last recorded behind (0: none)
    bool in_submit{false};              // inside cc_engine_add_firings_device: a synchronisation in there does not release the call being submitted
    bool cur_call_last{true};           // the sub-batch being submitted is the last one of its call
    int check_input_lifetime{0};        // option "check_input_lifetime": 1 checksum at submission and at release, 2 also poison on release
    struct InputRec
    {
        uint64_t seq;
        const void *xyz, *inten, *pose;
        size_t b_xyz, b_int, b_pose;
        unsigned long long sum;
    };
    std::vector<InputRec> live_inputs;  // (only with check_input_lifetime)
    unsigned long long* d_input_sum{nullptr};
    unsigned long long* h_input_sum{nullptr};
    bool pipelined{false};        // last submitted batch used all three streams
    bool allow_pipeline{true};    // option "pipeline"
    bool publish_off_chain{true}; // option "publish_off_chain"
    int table_on_insert_chain{1}; // option "table_on_insert_chain": k_table 0 = in front of the segmentation chain, 1 = at the end of the insertion chain,
                                  // 2 = on a (high-priority) stream of its own between the two
    bool ego_on_insert_chain{false};  // option "ego_on_insert_chain": k_ego (needs the caller's poses only) behind k_table instead of in front of k_seg_pre
    // CC_HOST_PROF=1: where the host's time goes inside a pipelined call (seconds, summed; printed by cc_engine_destroy)
    bool host_prof{false};
    double hp_pre{0}, hp_gate{0}, hp_post{0}, hp_entry{0};
    long long hp_calls{0};
    bool debug_no_assoc_fallback{false}; // option "debug_no_assoc_fallback": timing experiments only (results are wrong wherever k_assocb stops)
    bool parallel_insert_multi{true}; // option "parallel_insert" = 1: k_insert_multi behind / instead of k_insert_par; 2: k_insert_par only
    bool parallel_insert{true};   // option "parallel_insert": k_insert_par takes the single-column-firing head of every batch
    // low-latency path of cc_engine_add_firings for small calls: one captured hipGraph per (stream, n), pinned staging
    struct SmallGraph
    {
        int stream;
        int64_t n;
        int record;
        hipGraphExec_t exec;
    };
    std::vector<SmallGraph> small_graphs;
    unsigned char* h_small{nullptr}; // pinned: packed xyz | intensity | poses of up to SMALL_MAX firings
    unsigned char* d_small{nullptr};
    unsigned long long* h_small_seq{nullptr}; // pinned: calls whose results k_publish has mirrored (the host spins on it)
    unsigned long long* d_small_seq{nullptr}; // device: that number + a block counter
    unsigned long long small_seq_expected{0};
    unsigned long long small_tail_launches{0}; // small calls whose serial fall-backs the host launched behind k_small_all
    // ---- the resident single-stream kernel (option "resident", cc_k_publish.h: k_resident) ----
    // ---- the call's column views in the mirror (cc_k_publish.h: HostMirror::view) ----
    long long* h_small_view_hdr{nullptr};  // pinned: [0] call number, [1] columns, [2 .. 2 + MV_COLS) their global indices
    char* h_small_view{nullptr};           // pinned: view_layout(.., MV_COLS * rows)
    bool small_view_ok{false};             // the last small call's views are what the planes hold (nothing has changed the stream since)
    int small_view_stream{-1};
    bool mirror_views{true};               // option "mirror_views"
    unsigned long long view_hits{0}, view_misses{0};
    bool resident_opt{false};              // option "resident"
    bool res_running{false};               // k_resident sits on `stream`: nothing else may be enqueued there before stop_resident()
    cck::ResidentCtl* h_res_ctl{nullptr};  // pinned: doorbell, stop flag, exit reason
    int res_idle_ms{20};                   // option "resident_idle_ms": the kernel's watchdog (it leaves by itself after that long without a call)
    unsigned long long res_launches{0}, res_calls{0};
    StreamState* h_small_state{nullptr}; // pinned
    cc_event* h_small_events{nullptr};   // pinned
    bool allow_graphs{true};            // option "graphs"
    int scan_store_fin{-1};             // option "scan_store_fin": Planes::sc_fin written by the window scan and read by the serial kernels 1 always / 0 never (they recompute) / -1 per launch
    int scan_split{2};                  // option "scan_split": the packed window scan hands long scans to k_scan2_long (cc_k_scan.h); 2 = while there are many
    bool split_on{false};               // ... the automatic mode's current choice, from the counters k_scan2_epi leaves in d_bail_count[1 .. 2]
    unsigned split_cols_seen{0}, split_rec_seen{0}, split_probe{0};
    int scan_packed{-1};                // option "scan_packed": 1 = k_scan2 (active points packed into the lanes), 0 = k_scan (rows as lanes, lock
                                        // step), -1 (default) = k_scan2 for sensors with more than 64 rows (measured: S128 3.7 -> 2.0 ms per batch)
                                        // and, at up to 64 rows, for launches of more than 192 streams (there the step follows the number of vector
                                        // instructions, of which k_scan2 issues 0.65 x: + 3 %; below, the lock-step kernel's shorter launch wins 1 - 2 %)
    int assoc_waves{3};                 // option "assoc_waves": 1 = k_assoc_lds, 3 / 4 = k_assoc3 (2, the retired two-wavefront kernel, selects k_assoc3)
                                        // without / with its links wavefront
    bool assoc_batch{true};             // option "assoc_batch": k_assocb in front of the serial association kernels
    int assoc_rounds{0};                // option "assoc_rounds": (k_assocb, k_assoc3) pairs per batch; all but the last serial launch are limited.
                                        // 0 (default) = adaptive: one pair while no launch of k_assocb has had to stop lately, three for the 16
                                        // batches after one did (an empty launch of the serial kernel costs ~0.1 ms of chain time at 256 streams)
    bool assoc_waves_auto{true};        // ... 0 (default): k_assoc3, links wavefront while a launch has at most 256 streams
    bool assoc_pending[4]{false, false, false, false};
    std::vector<void*> allocations;
    // the chains of the last batch behind its insertion, not launched yet (launch_batch). With the lazy gate the closure first WAITS for that
    // insertion and reads its counters; `redo` (may be null) launches the next batch's insertion again when those counters say that the one
    // enqueued ahead of them was turned into a no-op
    std::function<int(const std::function<int()>*)> deferred_tail;
    int lazy_gate_max_streams{40};                       // option "lazy_gate": launches of at most this many streams (0: never) also enqueue the NEXT batch's insertion before they read this
                                                         // one's counters (32 streams + 6 %; at 64 the chains behind the gate start later than they should: - 2 %)
    int lazy_gate_from_streams{80};                      // option "lazy_gate_from": ... and launches of at least this many streams (0: none). Same-box alternations over 40 steps: 96 streams + 8 .. + 10 %,
                                                         // 128 / 160 / 192 + 0 .. + 2 %, 256 + 2 .. + 5 % (20 steps from an empty pipeline: + 0.5 .. + 0.9 %), 384 + 2 .. + 4 %; 48 streams - 1 %, 64 streams +- 0 (and - 2 % before the round's last changes)
    bool lazy_ok{true};                                  // (switched off for an engine whose streams keep needing the other insertion kernels)
    bool lazy_pending{false};                            // deferred_tail is such a closure: its batch's counters have not been read yet
    const int* lazy_prev_left{nullptr};                  // ... and this is where they are (device)
    int lazy_miss{0};
    int lazy_clean{0};                                   // batches in the steady shape seen by the plain gate since lazy_ok went off (eight re-arm it)
    uint64_t lazy_batches{0}, lazy_redone{0};            // cc_engine_gate_counters: insertions enqueued ahead of the previous batch's counters / launched a second time
    hipEvent_t ev_gate[4]{};
    int num_cus{256};                                    // compute units of the device
    int insert_lds_pad_kb{0};                            // option "insert_lds_pad" (experiment, default 0): launches of at most one block per compute unit ask for this much unused LDS per insertion
                                                         // block, so that no two of them share a CU. Measured with 81 KB at 256 streams: - 3.5 % (128 rows), - 6 % (64 rows): the blocks of the
                                                         // other chains' kernels lose the room, and the block times of k_insert_multi did not get shorter (the spread was not CU sharing)
    int defer_tail_max_streams{96};                      // option "defer_tail_max_streams": launches of at most this many streams defer them (0: never)
    bool streams_pooled{false};                          // the seven streams come from (and return to) the process-wide set cache
    bool slab_planning{false};                           // alloc_plane only records (field, offset): allocate() makes ONE hipMalloc of the total
    size_t slab_bytes{0};
    std::vector<std::pair<void**, size_t>> slab_plan;
    std::string error;
    // staging for the single-stream host path
    float* d_stage_xyz{nullptr};
    uint8_t* d_stage_int{nullptr};
    double* d_stage_pose{nullptr};
    int64_t stage_capacity{0};
    // staging of k_prep: [streams in launch][n][rows]
    size_t prep_capacity{0};
    // staging for cc_engine_read_columns
    void* d_view{nullptr};
    size_t view_bytes{0};
    double* d_ego[4]{nullptr, nullptr, nullptr, nullptr}; // k_ego output per batch-descriptor slot: [streams in launch][n][12]
    size_t ego_capacity{0};
    bool small_graphs_stale{false}; // a buffer a captured graph points at was re-allocated
    bool idle{false}; // nothing has been enqueued on any of the engine's HIP streams since they were last synchronised
    std::vector<StreamState> state_cache; // per stream: the state as the last small (graph) call copied it back, if still current
    std::vector<char> state_cached;
    char* d_gather{nullptr}; // scratch of cc_engine_gather_cluster_points
    size_t gather_bytes{0};
    char* h_view{nullptr}; // pinned mirror of d_view: one D2H copy per read, the fields are split on the host
    size_t h_view_bytes{0};
    std::vector<std::vector<cc_event>> pending_events; // per stream, drained from the device after each batch
    std::vector<std::vector<int64_t>> pending_links;   // per stream: (root gcol, root row, root gcol, root row) per logged tree link
    // pending continuation of the last device batch (kernel stopped early for some stream)
    const float* last_xyz{nullptr};
    const uint8_t* last_int{nullptr};
    const double* last_pose{nullptr};
    int64_t last_n{0};
    int64_t cur_ntotal{0}, cur_f0{0}; // the open batch is firings [cur_f0, cur_f0 + last_n) of buffers holding cur_ntotal per stream
    int64_t sub_batch{0};             // option "sub_batch": firings per pipelined sub-batch of a device call (0 = whole call)
    bool input_on_engine_stream{false}; // the producer of the device input buffers was enqueued on `stream`: order the preparation chain after it
    int last_first{0}, last_count{0};
    bool batch_open{false};
    // optional per-kernel timing with HIP events on the engine's stream (bench.py roofline leg)
    bool timing{false};
    int timing_every{1};       // option "timing_every": with timing on, the ten events bracket every n-th pass only (three event records in a row on the
                               // association stream are 15 - 25 us of its chain: 3.5 - 5 % of a step at 32 - 64 streams, 1 % at 256)
    uint64_t timing_pass{0};   // passes seen while timing was on (sampled or not)
    double prep_ahead_ms{0};   // k_prep launches that ran ahead of their pass (own event pairs, never sampled)
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used{0};
    double kernel_ms[7]{0, 0, 0, 0, 0, 0, 0}; // prep, insert+table, segment, scan, assoc_lds, assoc_global, publish
    uint64_t kernel_launches{0};
};

namespace
{

constexpr size_t CC_SLAB_ALIGN = 2u << 20;

#define CC_HIP_CHECK(e, call)                                                                                      \
    do                                                                                                             \
    {                                                                                                              \
        hipError_t _err = (call);                                                                                  \
        if (_err != hipSuccess)                                                                                    \
        {                                                                                                          \
            (e)->error = std::string(#call) + ": " + hipGetErrorString(_err);                                      \
            return CC_ERR_HIP;                                                                                     \
        }                                                                                                          \
    } while (0)

// One slab for the planes of an engine (allocate()): the planes are carved out of a single hipMalloc at 2 MiB boundaries, so that the driver
// can back them with its largest page-table fragments whatever the allocator's free lists look like by then — sixty separate allocations
// made after other engines of the process had come and gone ran the insertion kernel 1.3 - 1.6 x slower than the first engine's.
template<class T>
int alloc_plane(cc_engine* e, T** out, size_t count)
{
    if (e->slab_planning)
    {
        const size_t bytes = (count * sizeof(T) + CC_SLAB_ALIGN - 1) / CC_SLAB_ALIGN * CC_SLAB_ALIGN;
        e->slab_plan.push_back({(void**) out, e->slab_bytes});
        e->slab_bytes += bytes;
        return CC_OK;
    }
    void* p = nullptr;
    hipError_t err = hipMalloc(&p, count * sizeof(T));
    if (err != hipSuccess)
    {
        e->error = std::string("hipMalloc: ") + hipGetErrorString(err);
        return CC_ERR_HIP;
    }
    e->allocations.push_back(p);
    *out = (T*) p;
    return CC_OK;
}

int validate(cc_engine* e, const cc_config* cfg, int num_rows, int num_streams)
{
    if (!cfg || num_rows < 1 || num_rows > WAVE * MAX_ROWS_PER_LANE || num_streams < 1 || cfg->num_columns < 4 ||
        cfg->cluster_point_trees_every_nth_column < 1)
    {
        if (e)
            e->error = "invalid argument (num_rows must be 1..128, num_columns >= 4, every_nth_column >= 1)";
        return CC_ERR_INVALID_ARGUMENT;
    }
    return CC_OK;
}

void fill_geometry(cc_engine* e, int num_rows)
{
    Geometry& g = e->g;
    g.num_rows = num_rows;
    g.num_columns = e->cfg.num_columns;
    g.az_width = static_cast<float>((2 * M_PI)) / static_cast<float>(g.num_columns); // cc.cpp:16
    g.ring_cols = g.num_columns * 10;                                                 // cc.cpp:17
    g.cells = (int64_t) g.ring_cols * num_rows;
    g.max_distance_squared = e->cfg.max_distance * e->cfg.max_distance; // cc.cpp:80
    g.limit_columns = 2 * g.num_columns;
    g.tab_tiles = g.ring_cols / 64 + 2; // (a pass never emits more than a ring of columns)
    if (g.lds_tree_limit <= 0 || g.lds_tree_limit > TREE_SLOTS)
        g.lds_tree_limit = TREE_SLOTS;
    if (g.sl_cap <= 0 || g.sl_cap > cck::SL_CAP)
        g.sl_cap = cck::SL_CAP;
    if (g.scan_cap <= 0)
        g.scan_cap = cck::SCAN_CAP;
    g.scan_stores_fin = 0; // (set per launch: launch_batch)
}

int free_all(cc_engine* e)
{
    for (void* p : e->allocations)
        (void) hipFree(p);
    e->allocations.clear();
    e->d_stage_xyz = nullptr;
    e->d_stage_int = nullptr;
    e->d_stage_pose = nullptr;
    e->stage_capacity = 0;
    e->d_view = nullptr;
    e->view_bytes = 0;
    e->prep_capacity = 0;
    e->ego_capacity = 0;
    e->d_gather = nullptr;
    e->gather_bytes = 0;
    e->d_small = nullptr;
    e->d_small_seq = nullptr; // (freed with the allocations above; the pinned counter below goes with it)
    e->d_input_sum = nullptr;
    if (e->h_res_ctl)
        (void) hipHostFree(e->h_res_ctl);
    e->h_res_ctl = nullptr;
    if (e->h_small_view_hdr)
        (void) hipHostFree(e->h_small_view_hdr);
    if (e->h_small_view)
        (void) hipHostFree(e->h_small_view);
    e->h_small_view_hdr = nullptr;
    e->h_small_view = nullptr;
    e->small_view_ok = false;
    if (e->h_input_sum)
        (void) hipHostFree(e->h_input_sum);
    e->h_input_sum = nullptr;
    e->small_seq_expected = 0;
    // the pinned staging of the small-call path is sized for the row count it was created with
    if (e->h_small)
    {
        (void) hipHostFree(e->h_small);
        (void) hipHostFree(e->h_small_state);
        (void) hipHostFree(e->h_small_events);
        if (e->h_small_seq)
            (void) hipHostFree(e->h_small_seq);
        e->h_small_seq = nullptr;
        e->h_small = nullptr;
        e->h_small_state = nullptr;
        e->h_small_events = nullptr;
    }
    return CC_OK;
}

int allocate(cc_engine* e)
{
    const Geometry& g = e->g;
    const size_t S = (size_t) g.num_streams;
    const size_t C = S * (size_t) g.cells;
    const size_t L = S * (size_t) g.ring_cols;
    const size_t T = S * (size_t) g.tree_capacity;
    Planes& P = e->P;
    int rc = CC_OK;
    e->slab_planning = getenv("CC_NO_SLAB") == nullptr;
    e->slab_bytes = 0;
    e->slab_plan.clear();
#define A(field, count)                                \
    if ((rc = alloc_plane(e, &P.field, (count))) != 0) \
        return rc;
    A(dist, C) A(incl, C) A(incaz, C) A(gtag, C) A(src, C) A(inten, C);
    A(trig, L) A(colg, L) A(colminaz, L);
    A(ground, C) A(debug, C) A(ignored, C) A(root, C) A(id, C);
    A(t_fin, C) A(t_width, C) A(t_pts, C) A(t_uf, C) A(t_cid, C) A(t_pos, C) A(t_finished, C);
    A(ulist, T) A(ucomp, T) A(agg_fin, T) A(agg_min, T) A(agg_max, T) A(agg_pts, T) A(agg_cid, T) A(agg_first, T) A(agg_flag, T);
    A(events, S * (size_t) g.event_capacity);
    A(sc_parent, C) A(sc_nlinks, C) A(sc_links, C) A(sc_fin, C);
    A(sc_term, C) A(col_newfin, L) A(col_info, L) A(col_act, L) A(pk_meta, C) A(pk_fin, C) A(pk_lk, C);
    A(sg_x2, C) A(sg_uz, C) A(sg_w, C) A(sg_flags, C) A(sc_rec, C);
    A(curtab, S * (size_t) g.num_rows);
    A(tab_acc, S * (size_t) g.tab_tiles * (size_t) g.num_rows);
    A(par_off, S * (size_t) cck::IP_MAXF);
    A(tabc, (size_t) BATCH_SLOTS * S * (size_t) g.tab_tiles * (size_t) g.num_rows);
    A(sc_visits, C);
    A(link_log, S * (size_t) g.link_capacity);
    A(sl_ctl, S * 4) A(sl_cols, L);
    if ((rc = alloc_plane(e, (unsigned long long**) &P.sl_rec, S * (size_t) cck::SL_CAP * sizeof(cck::ScanLongRec) / sizeof(unsigned long long))) != 0)
        return rc;
#undef A
    if ((rc = alloc_plane(e, &e->d_states, S)) != 0)
        return rc;
    if ((rc = alloc_plane(e, &e->d_par_left, 16)) != 0) // (2 gate counters per batch slot; [8 + slot]: the in-kernel gate's count of streams that are through)
        return rc;
    if ((rc = alloc_plane(e, &e->d_bail_count, 4)) != 0) // [0] stops of k_assocb, [1] long-scan records, [2] columns scanned with the long scans apart
        return rc;
    if ((rc = alloc_plane(e, &e->d_remaining, 1)) != 0)
        return rc;
    const bool slab = e->slab_planning;
    if (e->slab_planning)
    {
        e->slab_planning = false;
        char* base = nullptr;
        if ((rc = alloc_plane(e, &base, e->slab_bytes)) != 0)
            return rc;
        for (auto& f : e->slab_plan)
            *f.first = base + f.second;
        e->slab_plan.clear();
    }
    (void) slab;
    // (counters the kernels only ever add to: recycled device memory is not zero)
    CC_HIP_CHECK(e, hipMemset(e->d_bail_count, 0, 4 * sizeof(int)));
    CC_HIP_CHECK(e, hipMemset(e->d_par_left, 0, 16 * sizeof(int)));
    CC_HIP_CHECK(e, hipMemset(e->d_remaining, 0, sizeof(int)));
    return CC_OK;
}

// reset(num_rows) of every stream, cc.cpp:11-64. `keep_table`: std::vector::resize keeps the old
// sc_inclination_angles_between_lasers_ values when the size does not change (cc.cpp:46).
int reset_state(cc_engine* e, bool keep_table)
{
    e->small_view_ok = false;
    std::fill(e->state_cached.begin(), e->state_cached.end(), 0);
    const Geometry& g = e->g;
    const size_t S = (size_t) g.num_streams;
    const size_t C = S * (size_t) g.cells;
    Planes& P = e->P;
    // cleared cell: distance = inclination = NaN (0xFF bytes), global column index = -1 (tag 0) (cc.cpp:1110-1119)
    CC_HIP_CHECK(e, hipMemsetAsync(P.dist, 0xFF, C * sizeof(float), e->stream));
    CC_HIP_CHECK(e, hipMemsetAsync(P.incl, 0xFF, C * sizeof(float), e->stream));
    CC_HIP_CHECK(e, hipMemsetAsync(P.gtag, 0, C * sizeof(uint16_t), e->stream));
    CC_HIP_CHECK(e, hipMemsetAsync(P.id, 0, C * sizeof(uint32_t), e->stream));
    CC_HIP_CHECK(e, hipMemsetAsync(P.ground, CC_GP_UNKNOWN, C, e->stream));
    CC_HIP_CHECK(e, hipMemsetAsync(P.debug, CC_DBG_WHITE, C, e->stream));
    CC_HIP_CHECK(e, hipMemsetAsync(P.ignored, 0, C, e->stream));
    CC_HIP_CHECK(e, hipMemsetAsync(P.root, 0xFF, C * sizeof(int32_t), e->stream));
    CC_HIP_CHECK(e, hipMemsetAsync(P.tab_acc, 0, S * (size_t) g.tab_tiles * (size_t) g.num_rows * sizeof(unsigned long long), e->stream));
    CC_HIP_CHECK(e, hipMemsetAsync(P.sl_ctl, 0, S * 4 * sizeof(int32_t), e->stream));
    if (!keep_table)
        CC_HIP_CHECK(e, hipMemsetAsync(P.curtab, 0xFF, S * (size_t) g.num_rows * sizeof(float), e->stream));
    std::vector<StreamState> init(S);
    for (auto& st : init)
    {
        memset(&st, 0, sizeof(st));
        st.prev_rearmost = 0;
        st.prev_foremost = -1;
        st.first_unfinished = -1;
        st.ring_start = -1;
        st.ring_end = -1;
        st.first_column = -1;
        st.clear_done = -1;
        st.first_unpublished = -1;
        st.cluster_counter = 1;
        st.min_required = 0;
        st.finish_lower_bound = std::numeric_limits<double>::max();
        st.last_round_min_az = -1.0; // Point::visited_at_continuous_azimuth_angle{-1.} cc.hpp:158
        st.overrun_col = std::numeric_limits<int64_t>::max();
        st.n_links = 0;
        for (auto& d : st.batch)
            d.seg_begin = d.seg_end = d.acp_next = d.pub_begin = d.pub_end = -1, d.mode = 0, d.fused = 0;
        st.assoc_mode = e->cfg.max_steps_in_row > WIN_COLS - 2 ? 1 : 0;
    }
    CC_HIP_CHECK(e, hipMemcpyAsync(e->d_states, init.data(), S * sizeof(StreamState), hipMemcpyHostToDevice, e->stream));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
    for (auto& v : e->pending_events)
        v.clear();
    for (auto& v : e->pending_links)
        v.clear();
    e->batch_open = false;
    for (bool& b : e->assoc_pending)
        b = false;
    // the lazy gate starts afresh with every epoch: a stream's first batches are never in the steady shape, and an engine reused for
    // another sequence must not inherit "two misses in a row" from the previous one (throughput would depend on history)
    e->deferred_tail = nullptr;
    e->lazy_pending = false;
    e->lazy_prev_left = nullptr;
    e->lazy_miss = 0;
    e->lazy_clean = 0;
    e->lazy_ok = true;
    return CC_OK;
}

int ensure_prep(cc_engine* e, size_t points)
{
    if (e->prep_capacity >= points)
        return CC_OK;
    int rc;
    // old blocks stay in `allocations` until the engine is destroyed or re-shaped (growth is rare: batch sizes repeat).
    // Two buffers: the points of batch b + 1 are prepared while batch b is still being inserted.
    for (auto& q : e->pp)
        if ((rc = alloc_plane(e, &q.x, points)) || (rc = alloc_plane(e, &q.y, points)) || (rc = alloc_plane(e, &q.z, points)) ||
            (rc = alloc_plane(e, &q.dist, points)) || (rc = alloc_plane(e, &q.incl, points)) ||
            (rc = alloc_plane(e, &q.incaz, points)) || (rc = alloc_plane(e, &q.cir, points)))
            return rc;
    e->prep_capacity = points;
    e->small_graphs_stale = true; // captured small-call graphs bake the old staging pointers
    return CC_OK;
}

int take_timing_events(cc_engine* e, hipEvent_t* ev)
{
    for (int i = 0; i < 10; i++)
    {
        if (e->ev_used == e->ev_pool.size())
        {
            hipEvent_t x;
            CC_HIP_CHECK(e, hipEventCreate(&x));
            e->ev_pool.push_back(x);
        }
        ev[i] = e->ev_pool[e->ev_used++];
    }
    return CC_OK;
}

static Planes planes_with_prep(const cc_engine* e, int buf)
{
    Planes P = e->P;
    const auto& q = e->pp[buf];
    P.pp_x = q.x;
    P.pp_y = q.y;
    P.pp_z = q.z;
    P.pp_dist = q.dist;
    P.pp_incl = q.incl;
    P.pp_incaz = q.incaz;
    P.pp_cir = q.cir;
    return P;
}

// k_prep of a batch on stream `sp` into staging buffer `buf` (the per-point part of insertion does not depend on engine state)
int launch_prep(cc_engine* e, int count, int64_t n, const float* d_xyz, const double* d_pose, int buf, hipStream_t sp, int64_t n_total,
                int64_t f0, bool skip_inserted = false, int first_stream = 0)
{
    const size_t points = (size_t) count * (size_t) n * e->g.num_rows;
    int rcp = ensure_prep(e, points);
    if (rcp)
        return rcp;
    const Planes P = planes_with_prep(e, buf);
    const size_t per_stream = (size_t) n * e->g.num_rows;
    hipLaunchKernelGGL(cck::k_prep, dim3((unsigned) ((per_stream + cck::PREP_POINTS_PER_BLOCK - 1) / cck::PREP_POINTS_PER_BLOCK), (unsigned) count), dim3(256), 0, sp,
                       e->g, e->cfg, P,
                       d_xyz, d_pose, (long long) n, (long long) n_total, (long long) f0, skip_inserted ? (const StreamState*) e->d_states : nullptr,
                       first_stream);
    return CC_OK;
}

// calls of a few firings on ONE stream outside the pipeline (cc_engine_add_firings: the per-column latency path) run everything in front of the window
// scan in one kernel
// The resident single-stream kernel (k_resident) occupies `stream` until it is told to leave: every entry point that enqueues work there, writes
// engine state from the host or changes what the kernel's arguments bake in calls this first. Read-only queries (cc_engine_read_columns,
// cc_engine_gather_cluster_points, drains of the mirrored events) do not: they run on query_stream() beside the idling kernel, whose writes are
// released to the device behind every call (mirror_results' system-scope fence).
static int stop_resident(cc_engine* e)
{
    if (!e->res_running)
        return CC_OK;
    __atomic_store_n(&e->h_res_ctl->stop, 1ull, __ATOMIC_RELEASE);
    const hipError_t err = hipStreamSynchronize(e->stream);
    e->res_calls += __atomic_load_n(&e->h_res_ctl->calls, __ATOMIC_ACQUIRE);
    e->res_running = false;
    __atomic_store_n(&e->h_res_ctl->stop, 0ull, __ATOMIC_RELEASE);
    if (err != hipSuccess)
    {
        e->error = std::string("stopping the resident kernel: ") + hipGetErrorString(err);
        return CC_ERR_HIP;
    }
    return CC_OK;
}

static hipStream_t query_stream(const cc_engine* e)
{
    return e->res_running ? e->stream2 : e->stream;
}

// Launch what launch_batch held back (below). Called before anything waits for, reads or re-orders the engine's streams.
static int flush_deferred(cc_engine* e, const std::function<int()>* redo = nullptr)
{
    if (!e->deferred_tail)
        return CC_OK;
    std::function<int(const std::function<int()>*)> f = std::move(e->deferred_tail);
    e->deferred_tail = nullptr;
    e->lazy_pending = false;
    return f(redo);
}

// The lazy gate (few streams: the insertion chain is what a step waits for, and between two of its kernels the GPU idled for as long as the host
// needs to notice the end of one and launch the next — 50 - 65 us of a 410 us step at 32 streams): a call enqueues its insertion and returns; the
// NEXT call enqueues ITS insertion first and only then waits for the previous one's counters and launches the chains behind it. The kernels of the
// insertion enqueued ahead read those counters themselves and do nothing if the previous batch is not complete (cc_k_insert.h: prev_left); the host
// then launches what the previous batch still needs and the insertion again. Same conditions as the deferred tail, which it extends.
static bool lazy_many_streams(const cc_engine* e, int count)
{
    return e->lazy_gate_from_streams > 0 && count >= e->lazy_gate_from_streams;
}

static bool lazy_eligible(const cc_engine* e, int count, int64_t n, bool pipeline, bool prepared)
{
    const int rpl = (e->g.num_rows + WAVE - 1) / WAVE;
    return pipeline && !prepared && (count <= e->lazy_gate_max_streams || lazy_many_streams(e, count)) && e->lazy_ok && e->parallel_insert && n >= 64 && n <= cck::IP_MAXF && rpl == 1 && e->fuse_front &&
           e->skip_idle_fallbacks && !e->capturing && e->defer_tail_max_streams > 0 && (count <= e->defer_tail_max_streams || lazy_many_streams(e, count)) && !e->host_prof &&
           !e->input_on_engine_stream && e->pipeline_depth >= 1;
}

// unused dynamic LDS that keeps a second block of the same kernel off the compute unit (see insert_lds_pad_kb)
static unsigned insert_lds_pad(const cc_engine* e, int blocks, size_t static_bytes)
{
    const size_t want = (size_t) e->insert_lds_pad_kb * 1024;
    return (e->insert_lds_pad_kb > 0 && blocks <= e->num_cus && want > static_bytes) ? (unsigned) (want - static_bytes) : 0u;
}

static bool use_small_front(const cc_engine* e, int count, int64_t n, bool pipeline)
{
    return e->small_front && !pipeline && count == 1 && n <= e->seg_small_max && n < 64 && e->g.num_rows <= WAVE;
}

// Begin a batch: zero the per-stream firing cursors and the early-stop counter; fix how far clearing may go.
__global__ void k_begin_batch(StreamState* states, int first_stream, int count, int* remaining, int unlimited_clear, int slot,
                              const int* __restrict__ prev_left = nullptr, int* __restrict__ gate_left = nullptr)
{
    if (prev_left && (prev_left[0] | prev_left[1]) != 0)
        return; // (the lazy gate: the previous batch's insertion is not complete — this launch must not have happened, cc_k_insert.h: k_insert_par)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && gate_left)
        gate_left[0] = gate_left[1] = 0; // (the counters k_insert_par leaves for the host's gate: zeroed here instead of by a memset node between two kernels)
    if (i < count)
    {
        states[first_stream + i].cursor = 0;
        states[first_stream + i].par_bad = 0x7fffffff;
        states[first_stream + i].par_upto = -1;
        states[first_stream + i].par_clear_done = -1;
        states[first_stream + i].pre_seg_begin = 0; // (k_insert2 clears it when it closes a batch; with skip_idle_fallbacks it may not have run)
        states[first_stream + i].n_events = 0; // every event of the previous call has been collected
        states[first_stream + i].n_links = 0;
        // (the descriptor slot is four batches old: whoever closes this batch as fused sets the flag again; nothing else may find it set)
        states[first_stream + i].batch[slot].fused = 0;
        states[first_stream + i].clear_allowed = unlimited_clear ? 0x7fffffffffffffffll : states[first_stream + i].ring_start;
    }
    if (i == 0)
        *remaining = 0;
}

// what the host reads behind a batch's insertion (the gate), written into pinned memory by a kernel: three 4-byte copies through the copy
// engine between the kernels of the insertion chain cost more than the kernels' dispatch
__global__ void k_gate_out(const int* __restrict__ left, const int* __restrict__ bail_count, const int* __restrict__ remaining, int* __restrict__ h_left,
                           int* __restrict__ h_bail_count, int* __restrict__ h_remaining)
{
    if (threadIdx.x == 0)
    {
        if (h_left)
        {
            h_left[0] = left[0];
            h_left[1] = left[1];
        }
        if (h_bail_count)
        {
            h_bail_count[0] = bail_count[0];
            h_bail_count[1] = bail_count[1]; // (long-scan statistics of the window scan: cc_k_scan.h, k_scan2_epi)
            h_bail_count[2] = bail_count[2];
        }
        if (h_remaining)
            *h_remaining = *remaining;
        __threadfence_system();
    }
}

int finish_batch(cc_engine* e);

// One pass over a batch: insertion on `si`, table + segmentation + window scan on `sb`, association + publish on `sa`
// (all three equal when not pipelined).
int launch_batch(cc_engine* e, int first_stream, int count, int64_t n, const float* d_xyz, const uint8_t* d_int,
                 const double* d_pose, bool first_pass, int slot, hipStream_t si, hipStream_t sb, hipStream_t sa,
                 hipStream_t sc = nullptr, hipStream_t sp = nullptr, bool prep_done = false)
{
    e->idle = false;
    {
        const size_t need = (size_t) count * (size_t) n;
        if (e->ego_capacity < need)
        {
            // (old blocks stay in `allocations`; captured small-call graphs hold the old pointers and are dropped)
            const size_t cap = need < 4096 ? 4096 : need;
            for (int i = 0; i < 4; i++)
            {
                int rce = alloc_plane(e, &e->d_ego[i], cap * cck::EGO_STRIDE);
                if (rce)
                    return rce;
            }
            e->ego_capacity = cap;
            e->small_graphs_stale = true;
        }
    }
    if (!sc)
        sc = sb; // window scan on the segmentation chain unless the four-stage pipeline gives it its own stream
    if (!sp)
        sp = si; // preparation on the insertion chain unless it runs ahead on its own stream
    const Geometry& g = e->g;
    const auto hp_t0 = std::chrono::steady_clock::now();
    auto hp_t1 = hp_t0;
    bool hp_gated = false;
    const int rpl = (g.num_rows + WAVE - 1) / WAVE;
    // an upper bound of the columns one pass can emit: the in-kernel limit plus half a rotation of one firing
    const long long max_cols = std::min<long long>((long long) g.limit_columns + g.num_columns, (long long) g.ring_cols);
    // grids are (streams, blocks): the stream index is the fast dimension so that one stream's blocks share an XCD (and its L2)
    dim3 seg_grid((unsigned) count, (unsigned) ((max_cols + 63) / 64));
    constexpr int NEV = 10;
    hipEvent_t ev[NEV] = {};
    const bool mark = e->timing && (e->timing_every <= 1 || e->timing_pass % (uint64_t) e->timing_every == 0);
    if (e->timing)
        e->timing_pass++;
    if (mark)
    {
        int rct = take_timing_events(e, ev);
        if (rct)
            return rct;
    }
    int k = 0;
#define CC_MARK(st_) \
    if (mark)        \
        CC_HIP_CHECK(e, hipEventRecord(ev[k++], st_));
    // ---- insertion chain -----------------------------------------------------------------------------------------
    // The head of the batch that has the single-column firing shape is inserted by all wavefronts of a block at once, straight from
    // the caller's buffers; preparation and the serial kernel then only see what is left (StreamState::cursor).
    const bool par = first_pass && !prep_done && e->parallel_insert && n >= 64; // (small calls are latency-bound: one kernel less)
    if (par)
        sp = si;
    CC_MARK(sp); // ev0
    // skip_idle_fallbacks: in steady state k_insert_par takes whole batches and the three kernels behind it (k_insert_multi, k_prep, k_insert2)
    // have nothing to do — but their blocks wait for free CUs next to the throughput kernels of the other chains, 0.2 - 0.4 ms of chain time per
    // batch. The host has to wait for the insertion chain before the next batch anyway, so it waits here, for k_insert_par alone, and launches the
    // others only if some stream's batch was not taken completely (the kernel then leaves the batch descriptor to k_insert2 as before).
    // (with the fused segmentation also outside the pipelined mode: the fused path needs the counters the gate reads)
    const bool gate = par && rpl == 1 && (si != sb || e->fuse_front) && e->skip_idle_fallbacks && n <= cck::IP_MAXF && !e->capturing;
    // Few streams: a step is as long as its insertion chain PLUS the host's launches of the other chains, because the host waits at the gate
    // before it launches them and the next batch's insertion only starts behind all of that. So the chains behind the gate (`tail` below) of a
    // batch that needs nothing more on the insertion stream are held back and launched by the NEXT call, after that call has enqueued its own
    // insertion and before it waits at its gate: the insertion kernels run back to back and the launches hide behind them. Anything that waits
    // for or reads results launches the held-back chains first (flush_deferred in sync_all).
    // (launches of many streams only defer together with the lazy gate: that pair is what was measured there)
    const bool may_defer = gate && first_pass && si != sb && si != sa && e->defer_tail_max_streams > 0 &&
                           (count <= e->defer_tail_max_streams || (lazy_many_streams(e, count) && lazy_eligible(e, count, n, true, prep_done)));
    if (!gate)
    {
        int rcf = flush_deferred(e);
        if (rcf)
            return rcf;
    }
    bool fallbacks = true;
    bool need_segpre = true; // some stream's batch is not closed as fused: k_table / k_seg_pre have work
    bool ego_done = false;
    // k_table -> k_seg_scan scratch of this batch-descriptor slot (up to BATCH_SLOTS batches are in flight)
    Planes Pt = e->P;
    Pt.tabc += (size_t) slot * (size_t) g.num_streams * (size_t) g.tab_tiles * (size_t) g.num_rows;
    // per-firing ego transforms of this batch (one buffer per descriptor slot: up to three batches are in flight)
    double* d_ego = e->d_ego[slot];
    const bool lazy = may_defer && lazy_eligible(e, count, n, true, prep_done);
    bool lazy_registered = false;
    int* const gate_left = e->d_par_left + 2 * slot; // (one pair of counters per batch descriptor slot: the lazy gate reads a batch's pair while the next batch runs)
    int* const gate_h_left = e->h_par_left + 2 * slot;
    if (par && rpl == 1) // (two rows per lane = sensors with per-laser azimuth offsets in practice: straight to k_insert_multi)
    {
        const bool fuse = gate && e->fuse_front;
        const double* ego_in = fuse ? (const double*) d_ego : (const double*) nullptr;
        int* left = gate ? gate_left : (int*) nullptr;
        const long long cur_ntotal = e->cur_ntotal, cur_f0 = e->cur_f0;
        // prev_left: the counters of the previous batch's insertion when this one is enqueued before the host has read them (lazy gate)
        const bool gate_zeroed = gate && first_pass && si != sb && !use_small_front(e, count, n, true); // (submit's k_begin_batch zeroed the counters)
        auto enqueue_insertion = [=](const int* prev_left, const bool with_remaining) -> int
        {
            if (fuse)
            {
                // the fused insertion needs the per-firing ego records: they only depend on the caller's poses (and the robot transform, which the
                // host writes between batches) — so not on the insertion chain, which is what a step waits for: on the preparation stream (idle while the
                // block-parallel insertion takes the batches), where they run beside the PREVIOUS batch's insertion; the insertion waits for the
                // event. The records' buffer belongs to the batch-descriptor slot: its last readers (segmentation chain of four batches ago) are
                // in front of that slot's publishing event.
                const bool off = e->ego_off_chain && si != sb && !e->capturing;
                hipStream_t se = off ? e->stream5 : si;
                if (off)
                    CC_HIP_CHECK(e, hipStreamWaitEvent(se, e->ev_assoc[slot], 0));
                hipLaunchKernelGGL(cck::k_ego, dim3((unsigned) ((n + 255) / 256), (unsigned) count), dim3(256), 0, se, (const StreamState*) e->d_states, first_stream,
                                   e->cfg, d_pose, (long long) n, cur_ntotal, cur_f0, d_ego);
                if (off)
                {
                    CC_HIP_CHECK(e, hipEventRecord(e->ev_ego[slot], se));
                    CC_HIP_CHECK(e, hipStreamWaitEvent(si, e->ev_ego[slot], 0));
                }
            }
            if (gate && !gate_zeroed)
                CC_HIP_CHECK(e, hipMemsetAsync(left, 0, 2 * sizeof(int), si));
            // few streams: the GPU is not full and the insertion chain is what a step waits for -> twice the wavefronts per block, and the firings of a
            // stream dealt to several blocks (k_insert_par_fin then finishes the stream's state)
            if (count <= e->insert_wide_max_streams)
            {
                // (round 4: with the segmentation fused in, a block of 8 wavefronts needs ~1.1 ms per 2200 firings by itself: up to 160 streams the
                // GPU has room for twice the wavefronts — 128 streams 11.3 -> 15.0 G points/s — above that it is full and they only get in each other's way)
                // (blocks per stream, same-box alternations over 40 steps: 4 up to 40 streams; 3 up to 64 — 48 streams 12.3 -> 12.9 - 13.0 G points/s and 64 streams
                // 14.0 - 14.2 -> 14.4 - 14.6 against 2 blocks, 4 blocks at 64 streams - 5 %; 2 up to 96 — at 80 streams 3 blocks are 5 - 8 % slower than 2)
                // A block of 16 wavefronts wants a compute unit it does not share with a block of k_assocb (one per stream, 16 wavefronts too): blocks x streams + streams <= 256
                // is where more blocks stop paying — 32 streams: 4 / 6 / 7 / 8 blocks 11.3 / 11.7 - 12.0 / 11.6 - 12.0 / 9.7 G points/s; 24 and 16 streams: 8 blocks + 1 .. + 3 % against 4;
                // 40 streams: 5 blocks - 3 .. - 5 % against 4 (the rule is not exact: measured points decide)
                const int nb = e->insert_split_blocks > 0 ? e->insert_split_blocks
                                                          : (count <= 24 ? 8 : (count <= 32 ? 6 : (count <= 40 ? 4 : (count <= 64 ? 3 : (count <= 96 ? 2 : 1)))));
                hipLaunchKernelGGL((cck::k_insert_par<1, 2 * cck::IP_WAVES>), dim3(count, nb), dim3(128 * cck::IP_WAVES), insert_lds_pad(e, count * nb, 20 * 1024), si, g, e->cfg, Pt, e->d_states,
                                   first_stream, d_xyz, d_int, d_pose, (long long) n, cur_ntotal, cur_f0, slot, left, ego_in, prev_left);
                if (nb > 1)
                    hipLaunchKernelGGL(cck::k_insert_par_fin<1>, dim3(count), dim3(256), 0, si, g, Pt, e->d_states, first_stream, d_xyz, (long long) n,
                                       cur_ntotal, cur_f0, slot, left, fuse ? 1 : 0, prev_left);
            }
            else if (e->insert_narrow_blocks > 0)
            {
                // (experiment: small blocks find room on a busy CU sooner than one block of 8 wavefronts)
                hipLaunchKernelGGL((cck::k_insert_par<1, 4>), dim3(count, e->insert_narrow_blocks), dim3(256), 0, si, g, e->cfg, Pt, e->d_states,
                                   first_stream, d_xyz, d_int, d_pose, (long long) n, cur_ntotal, cur_f0, slot, left, ego_in, prev_left);
                if (e->insert_narrow_blocks > 1)
                    hipLaunchKernelGGL(cck::k_insert_par_fin<1>, dim3(count), dim3(256), 0, si, g, Pt, e->d_states, first_stream, d_xyz, (long long) n,
                                       cur_ntotal, cur_f0, slot, left, fuse ? 1 : 0, prev_left);
            }
            else
                hipLaunchKernelGGL((cck::k_insert_par<1, cck::IP_WAVES>), dim3(count), dim3(64 * cck::IP_WAVES), insert_lds_pad(e, count, 20 * 1024), si, g, e->cfg, Pt, e->d_states,
                                   first_stream, d_xyz, d_int, d_pose, (long long) n, cur_ntotal, cur_f0, slot, left, ego_in, prev_left);
            if (gate)
            {
                // (the counter of k_assocb's stops rides along: as of whatever the association chain has finished by now — it only steers a heuristic;
                // with the lazy gate also the early-stop counter the held-back chains would have copied)
                hipLaunchKernelGGL(k_gate_out, dim3(1), dim3(64), 0, si, (const int*) left, (const int*) e->d_bail_count, (const int*) e->d_remaining, gate_h_left,
                                   e->h_bail_count, with_remaining ? e->h_remaining : (int*) nullptr);
            }
            return CC_OK;
        };
        if (fuse)
            ego_done = true;
        if (lazy)
        {
            // this batch's insertion goes out before the previous one's counters have been read; what the held-back chains need of the insertion
            // stream (the early-stop counter, the event the segmentation chain waits for) and the event the NEXT call waits for follow it
            auto enqueue_lazy = [=](const int* prev_left) -> int
            {
                int rci = enqueue_insertion(prev_left, true);
                if (rci)
                    return rci;
                CC_HIP_CHECK(e, hipEventRecord(e->ev_ins[slot], si));
                CC_HIP_CHECK(e, hipEventRecord(e->ev_gate[slot], si));
                return CC_OK;
            };
            int rcl = enqueue_lazy(e->lazy_pending ? e->lazy_prev_left : nullptr);
            if (rcl)
                return rcl;
            e->lazy_batches += e->lazy_pending ? 1 : 0;
            CC_MARK(sp); // ev1
            CC_MARK(si); // ev2
            // now the previous batch: its counters, the chains behind its insertion — or, if it needs the other insertion kernels, those first and
            // then this batch's insertion once more (the one above did nothing)
            const std::function<int()> redo = [=]() -> int
            {
                hipLaunchKernelGGL(k_begin_batch, dim3((count + 255) / 256), dim3(256), 0, si, e->d_states, first_stream, count, e->d_remaining, 1, slot,
                                   (const int*) nullptr, gate_left);
                return enqueue_lazy(nullptr);
            };
            int rcf = flush_deferred(e, &redo);
            if (rcf)
                return rcf;
            e->idle = false; // (the previous batch's closure may have gone through finish_batch / sync_all: this batch's insertion is in flight)
            fallbacks = false; // (as far as anybody knows: the closure registered below finds out)
            need_segpre = false;
            lazy_registered = true;
        }
        else
        {
            int rci = enqueue_insertion(nullptr, false);
            if (rci)
                return rci;
        }
        if (gate && !lazy)
        {
            {
                // (this batch's insertion is enqueued: now the chains of the previous batch that were held back)
                int rcf = flush_deferred(e);
                if (rcf)
                    return rcf;
            }
            const auto hp1 = std::chrono::steady_clock::now();
            CC_HIP_CHECK(e, hipStreamSynchronize(si));
            hp_t1 = std::chrono::steady_clock::now();
            if (e->host_prof)
            {
                e->hp_pre += std::chrono::duration<double>(hp1 - hp_t0).count();
                e->hp_gate += std::chrono::duration<double>(hp_t1 - hp1).count();
                hp_gated = true;
            }
            fallbacks = gate_h_left[0] != 0;
            need_segpre = gate_h_left[1] != 0;
            // an engine that lost the lazy gate (two misses in a row: start-up, sub-rotation batches) gets it back after eight batches in the
            // steady shape; one more miss then switches it off again at once
            if (!e->lazy_ok)
            {
                e->lazy_clean = (fallbacks || need_segpre) ? 0 : e->lazy_clean + 1;
                if (e->lazy_clean >= 8)
                    e->lazy_ok = true, e->lazy_miss = 1, e->lazy_clean = 0;
            }
        }
    }
    // multi-column firings (per-laser azimuth offsets) and whatever single-column head k_insert_par did not take: block-parallel as well,
    // with the per-row collision rule checked instead of assumed (option "parallel_insert" = 2 restricts this to the first kernel)
    // (above 64 rows it is the first insertion kernel, and the gate is here: k_prep and k_insert2<2> — 96 KB of LDS per block — stood 1.3 ms per batch in
    // the insertion chain of 256 VLS-128-shaped streams, the chain the host waits for, to find nothing to do)
    const bool gate2 = par && rpl > 1 && e->parallel_insert_multi && si != sb && e->skip_idle_fallbacks;
    if (par && e->parallel_insert_multi && fallbacks)
    {
        if (gate2)
            CC_HIP_CHECK(e, hipMemsetAsync(e->d_par_left, 0, sizeof(int), si));
        int* left2 = gate2 ? e->d_par_left : (int*) nullptr;
        if (rpl == 1)
            hipLaunchKernelGGL(cck::k_insert_multi<1>, dim3(count), dim3(64 * cck::IM_WAVES), 0, si, g, e->cfg, e->P, e->d_states, first_stream, d_xyz,
                               d_int, d_pose, (long long) n, (long long) e->cur_ntotal, (long long) e->cur_f0, slot, (int*) nullptr);
        else
            hipLaunchKernelGGL(cck::k_insert_multi<2>, dim3(count), dim3(64 * cck::IM_WAVES), 0, si, g, e->cfg, e->P, e->d_states, first_stream, d_xyz,
                               d_int, d_pose, (long long) n, (long long) e->cur_ntotal, (long long) e->cur_f0, slot, left2);
        if (gate2)
        {
            CC_HIP_CHECK(e, hipMemcpyAsync(e->h_par_left, e->d_par_left, sizeof(int), hipMemcpyDeviceToHost, si));
            if (e->h_bail_count)
                CC_HIP_CHECK(e, hipMemcpyAsync(e->h_bail_count, e->d_bail_count, 3 * sizeof(int), hipMemcpyDeviceToHost, si));
            const auto hp1 = std::chrono::steady_clock::now();
            CC_HIP_CHECK(e, hipStreamSynchronize(si));
            hp_t1 = std::chrono::steady_clock::now();
            if (e->host_prof)
            {
                e->hp_pre += std::chrono::duration<double>(hp1 - hp_t0).count();
                e->hp_gate += std::chrono::duration<double>(hp_t1 - hp1).count();
                hp_gated = true;
            }
            fallbacks = *e->h_par_left != 0;
        }
    }
    // what the held-back chains would still put on the insertion stream is put there now (time marks, the early-stop counter, the event the
    // segmentation chain waits for): the held-back part must not touch that stream, the next batch's insertion will be on it by then
    bool pre_done = lazy_registered;
    const bool defer = may_defer && !fallbacks && !need_segpre && !e->capture_mirror.state;
    if (defer && !lazy_registered)
    {
        CC_MARK(sp); // ev1
        CC_MARK(si); // ev2
        CC_HIP_CHECK(e, hipMemcpyAsync(e->h_remaining, e->d_remaining, sizeof(int), hipMemcpyDeviceToHost, si));
        CC_HIP_CHECK(e, hipEventRecord(e->ev_ins[slot], si));
        pre_done = true;
    }
    // the call this batch is the end of (0: a continuation pass, a sub-batch that is not its call's last, a call on the host path)
    const uint64_t rel_seq = (first_pass && e->in_submit && e->cur_call_last) ? e->call_seq : 0ull;
    // (the three arguments: >= 0 replaces what was known when the closure was made — the lazy gate learns them later)
    auto tail = [=](const int fb_now, const int seg_now, const int pre_now) mutable -> int
    {
        if (fb_now >= 0)
            fallbacks = fb_now != 0;
        if (seg_now >= 0)
            need_segpre = seg_now != 0;
        if (pre_now >= 0)
            pre_done = pre_now != 0;
        const bool small_front = first_pass && !prep_done && !par && use_small_front(e, count, n, si != sb);
        if (small_front)
        {
            int rcp = ensure_prep(e, (size_t) n * g.num_rows);
            if (rcp)
                return rcp;
            const Planes Pf = planes_with_prep(e, e->prep_buf);
            // the whole call in one launch where the results go to pinned memory (the captured graph of cc_engine_add_firings' small calls): the
            // serial fall-backs, needed once in a long while, are launched by the host when the kernel asks for them (add_firings_small)
            const bool small_all = e->small_all && e->capture_mirror.state != nullptr && e->capture_mirror.tail_req != nullptr && e->assoc_batch &&
                                   e->assoc_waves >= 2 && rpl == 1 && e->cfg.cluster_point_trees_every_nth_column == 1 && !e->debug_no_assoc_fallback;
            if (small_all)
            {
                hipLaunchKernelGGL(cck::k_small_all, dim3(1), dim3(cck::AB_THREADS), cck::insert2_lds_bytes(g.num_rows), si, g, e->cfg, Pf, e->d_states, first_stream,
                                   slot, d_xyz, d_int, d_pose, (long long) n, e->d_remaining, d_ego, e->d_bail_count, e->capture_mirror);
                CC_HIP_CHECK(e, hipGetLastError());
                return CC_OK;
            }
            hipLaunchKernelGGL(cck::k_small_front, dim3(1), dim3(256), cck::insert2_lds_bytes(g.num_rows), si, g, e->cfg, Pf, e->d_states, first_stream, slot, d_xyz,
                               d_int, d_pose, (long long) n, e->d_remaining, d_ego);
            fallbacks = false;
            need_segpre = false;
        }
        if (first_pass && !prep_done && fallbacks) // relaunch passes of the same batch reuse the staged points; a pipelined caller prepared ahead
        {
            int rcp = launch_prep(e, count, n, d_xyz, d_pose, e->prep_buf, sp, e->cur_ntotal, e->cur_f0, par, first_stream);
            if (rcp)
                return rcp;
        }
        if (!pre_done)
            CC_MARK(sp); // ev1: prep (with k_insert_par in front of it when that is on)
        if (sp != si)
        {
            CC_HIP_CHECK(e, hipEventRecord(e->ev_prep[slot], sp));
            CC_HIP_CHECK(e, hipStreamWaitEvent(si, e->ev_prep[slot], 0));
        }
        const Planes Pins = planes_with_prep(e, e->prep_buf);
        if (fallbacks)
        {
            const size_t lds = cck::insert2_lds_bytes(g.num_rows);
            if (rpl == 1)
                hipLaunchKernelGGL(cck::k_insert2<1>, dim3(count), dim3(128), lds, si, g, e->cfg, Pins, e->d_states, first_stream, slot,
                                   d_int, (long long) n, e->d_remaining, (long long) e->cur_ntotal, (long long) e->cur_f0);
            else
                hipLaunchKernelGGL(cck::k_insert2<2>, dim3(count), dim3(128), lds, si, g, e->cfg, Pins, e->d_states, first_stream, slot,
                                   d_int, (long long) n, e->d_remaining, (long long) e->cur_ntotal, (long long) e->cur_f0);
        }
        if (!pre_done)
            CC_MARK(si); // ev2: insert
        if (!pre_done && !e->capture_mirror.state) // (a small call's graph gets the counter through k_publish's mirror)
            CC_HIP_CHECK(e, hipMemcpyAsync(e->h_remaining, e->d_remaining, sizeof(int), hipMemcpyDeviceToHost, si));
        // k_table only needs what the insertion of this batch wrote. It is a latency-bound kernel (8 wavefronts per stream) that takes 0.8 ms
        // when it shares the GPU with the throughput kernels — on the segmentation chain, which is the longest of the three, that is a
        // third of the chain; at the end of the insertion chain, which has slack, it costs nothing.
        // calls of a few firings (the per-column latency path): ONE wavefront per stream segments the call's columns, rows as lanes (k_seg_small)
        const bool seg_small = !par && rpl == 1 && first_pass && n <= e->seg_small_max && !small_front;
        if (seg_small)
            need_segpre = false;
        // (with the fused front half k_insert_par reads and writes the running table `curtab` on the insertion chain: k_table of a batch that is not
        // fused has to run on that chain too, whatever the option says — elsewhere nothing would order it against the next batch's insertion)
        const int table_opt = (e->fuse_front && rpl == 1) ? 1 : e->table_on_insert_chain; // (above 64 rows nothing is fused: only k_table touches the running table)
        const bool table_early = si != sb && table_opt != 0;
        // (a stream of its own: the next batch's insertion does not queue behind it)
        hipStream_t st_table = (table_early && table_opt == 2 && !e->capturing) ? e->stream7 : si;
        if (table_early && need_segpre)
        {
            if (st_table != si)
            {
                CC_HIP_CHECK(e, hipEventRecord(e->ev_ins[slot], si));
                CC_HIP_CHECK(e, hipStreamWaitEvent(st_table, e->ev_ins[slot], 0));
            }
            if (rpl == 1)
                hipLaunchKernelGGL(cck::k_table<1>, dim3(count), dim3(64 * cck::TABLE_WAVES), 0, st_table, g, Pt, e->d_states, first_stream, slot);
            else
                hipLaunchKernelGGL(cck::k_table<2>, dim3(count), dim3(64 * cck::TABLE_WAVES), 0, st_table, g, Pt, e->d_states, first_stream, slot);
        }
        const bool ego_early = table_early && e->ego_on_insert_chain;
        if (ego_early && !ego_done && need_segpre)
            hipLaunchKernelGGL(cck::k_ego, dim3((unsigned) ((n + 255) / 256), (unsigned) count), dim3(256), 0, st_table, (const StreamState*) e->d_states, first_stream,
                               e->cfg, d_pose, (long long) n, (long long) e->cur_ntotal, (long long) e->cur_f0, d_ego);
        if (si != sb)
        {
            if (!pre_done)
                CC_HIP_CHECK(e, hipEventRecord(e->ev_ins[slot], st_table));
            CC_HIP_CHECK(e, hipStreamWaitEvent(sb, e->ev_ins[slot], 0));
        }
        // ---- table + segmentation + window-scan chain ------------------------------------------------------------
        CC_MARK(sb); // ev3: start of the second chain
        if (!table_early && need_segpre)
        {
            if (rpl == 1)
                hipLaunchKernelGGL(cck::k_table<1>, dim3(count), dim3(64 * cck::TABLE_WAVES), 0, sb, g, Pt, e->d_states, first_stream, slot);
            else
                hipLaunchKernelGGL(cck::k_table<2>, dim3(count), dim3(64 * cck::TABLE_WAVES), 0, sb, g, Pt, e->d_states, first_stream, slot);
        }
        if (!ego_early && !ego_done && need_segpre)
            hipLaunchKernelGGL(cck::k_ego, dim3((unsigned) ((n + 255) / 256), (unsigned) count), dim3(256), 0, sb, (const StreamState*) e->d_states, first_stream,
                           e->cfg, d_pose, (long long) n, (long long) e->cur_ntotal, (long long) e->cur_f0, d_ego);
        if (!need_segpre)
            ;
        else if (rpl == 1)
            hipLaunchKernelGGL(cck::k_seg_pre<1>, dim3((unsigned) count, cck::SEGPRE_BLOCKS), dim3(64), 0, sb, g, e->cfg, Pt, e->d_states,
                               first_stream, slot, d_pose, (long long) e->cur_ntotal, (long long) e->cur_f0, (const double*) d_ego, (long long) n);
        else
            hipLaunchKernelGGL(cck::k_seg_pre<2>, dim3((unsigned) count, cck::SEGPRE_BLOCKS), dim3(64), 0, sb, g, e->cfg, Pt, e->d_states,
                               first_stream, slot, d_pose, (long long) e->cur_ntotal, (long long) e->cur_f0, (const double*) d_ego, (long long) n);
        if (seg_small)
        {
            hipLaunchKernelGGL(cck::k_ego, dim3((unsigned) ((n + 255) / 256), (unsigned) count), dim3(256), 0, sb, (const StreamState*) e->d_states, first_stream,
                               e->cfg, d_pose, (long long) n, (long long) e->cur_ntotal, (long long) e->cur_f0, d_ego);
            hipLaunchKernelGGL(cck::k_seg_small, dim3((unsigned) count), dim3(64), 0, sb, g, e->cfg, e->P, e->d_states, first_stream, slot, d_pose,
                               (long long) e->cur_ntotal, (long long) e->cur_f0, (const double*) d_ego, (long long) n);
        }
        else if (!small_front)
        {
            const size_t lds = cck::seg_scan_lds_bytes(g.num_rows);
            hipLaunchKernelGGL(cck::k_seg_scan, seg_grid, dim3(64), lds, sb, g, e->cfg, Pt, e->d_states, first_stream, slot); // (Pt: this slot's table carries)
        }
        if (sc != sb)
        {
            CC_HIP_CHECK(e, hipEventRecord(e->ev_segscan[slot], sb));
            CC_HIP_CHECK(e, hipStreamWaitEvent(sc, e->ev_segscan[slot], 0));
        }
        CC_MARK(sc); // ev4: table + segment (start of the window scan)
        // ---- what the association chain of this batch will be (decided here: the window scan writes Planes::sc_fin only for the serial kernels) ----
        bool batch_assoc = e->assoc_batch && e->cfg.cluster_point_trees_every_nth_column == 1;
        // Streams on which k_assocb keeps stopping (vegetation: more trees born per group than it has lanes for) cost a batch more with it than
        // without: every stop is a (batch-parallel, serial) round, and a launch lasts as long as its slowest stream — the one that went serial.
        // While at least a quarter of a launch's streams stop per batch the serial kernels run alone; every ninth batch tries again.
        if (batch_assoc && e->assoc_rounds == 0 && e->h_bail_count && !e->capturing && count >= 8)
        {
            const int seen_now = *e->h_bail_count;
            if (e->chronic_skip > 0)
            {
                e->chronic_skip--;
                e->bail_seen = seen_now;
                batch_assoc = false;
                // (the batches that try again must not meet the two sweeping blocks the serial kernel runs as behind an idle k_assocb)
                if (e->chronic_skip == 0)
                    e->bail_cooldown = e->bail_cooldown_batches > 2 ? e->bail_cooldown_batches : 2;
            }
            else if ((seen_now - e->chronic_seen) * 4 >= count && e->chronic_probe)
                e->chronic_skip = 8;
            e->chronic_probe = batch_assoc; // (the counter read behind the NEXT batch tells what this one did)
            e->chronic_seen = seen_now;
        }
        int adaptive_rounds = 1;
        if (e->assoc_rounds == 0 && e->h_bail_count && !e->capturing)
        {
            const int seen = *e->h_bail_count; // (as of some earlier batch: a heuristic, not a condition of correctness)
            if (seen != e->bail_seen)
            {
                e->bail_seen = seen;
                e->bail_cooldown = e->bail_cooldown_batches;
            }
            if (e->bail_cooldown > 0)
            {
                e->bail_cooldown--;
                adaptive_rounds = 3;
            }
        }
        // The serial kernels read a point's finished_at contribution from Planes::sc_fin or recompute it (cc_k_base.h: cell_fin_of). Behind the
        // batch-parallel kernel they find nothing to do, and the scan saves the 8 bytes per cell; where they are expected to associate (the
        // batch-parallel kernel off, pinned rounds, stops lately) the scan stores them. A small call's front kernel has scanned with the engine's
        // geometry (never stored).
        Geometry gs = g;
        gs.scan_stores_fin = (!small_front && !(batch_assoc && e->assoc_rounds == 0 && adaptive_rounds == 1)) ? 1 : 0;
        if (e->scan_store_fin >= 0 && !small_front)
            gs.scan_stores_fin = e->scan_store_fin;
        const dim3 scan_grid((unsigned) count, cck::SCAN_BLOCKS);
        bool use_split = false;
        if (small_front)
            ; // (k_small_front has scanned the call's columns)
        // (65 - 128 rows: packed by default. The lock-step form with two rows per lane — scan_packed = 0 — shortens the scan's own launch, 3.0 -> 2.35 ms at
        // 256 x S128, but needs more vector instructions, and the step is bound by those: 11.7 -> 11.4 G points/s same-box)
        // (64 rows, end of round 4: with the insertion's uniform work on the scalar unit the step follows the vector-instruction count, and the packed
        // scan issues 0.65 x those of the lock-step one: + 3 % at 256 streams (same-box, 3 alternations: 16.22 -> 16.72 G points/s), - 1 ... - 2 % at
        // 32 - 128 streams where the GPU has room and the lock-step scan's shorter launch counts)
        else if ([&]() -> bool
                 {
                     // the long scans apart? scan_split 1: always (with the packed scan); 2 (default): while they are a large part of the scan's work. The
                     // visits k_scan2_long makes per column (of 64 rows) say so: vegetation ~150, the 128-row bench scene ~15, the street scene ~4. Where they
                     // are few the split costs chain time (two more launches whose blocks wait for wave slots, the longest single scan standing alone: street
                     // scene - 8 % at 256 streams, the 128-row scene - 2 %), on vegetation it is + 60 .. + 70 %. Every 32nd batch is scanned packed and with
                     // the split, which counts; the batches counted since the last look decide (on above 40 visits per column, off again below 20).
                     // On vegetation the packed scan with the split also beats the lock-step scan from 48 streams per launch (64 streams + 14 %, 128 + 38 %;
                     // 32 streams - 4 %), where the street scene wants the lock-step one up to 192.
                     const bool packed_default = e->scan_packed == 1 || (e->scan_packed < 0 && (rpl > 1 || count > 192));
                     use_split = false;
                     if (g.mirror_fields || e->scan_split == 0)
                         return packed_default;
                     if (e->scan_split == 1 || !e->h_bail_count || e->capturing)
                     {
                         use_split = packed_default && e->scan_split == 1;
                         return packed_default;
                     }
                     const unsigned vis = (unsigned) e->h_bail_count[1], cols = (unsigned) e->h_bail_count[2];
                     const unsigned dc = cols - e->split_cols_seen, dv = vis - e->split_rec_seen;
                     if (dc >= 1024u)
                     {
                         const double rate = (double) dv / ((double) dc * (double) rpl); // (per column of 64 rows)
                         e->split_on = e->split_on ? rate > 20.0 : rate > 40.0;
                         e->split_cols_seen = cols, e->split_rec_seen = vis;
                     }
                     const bool probe = (e->split_probe++ & 31u) == 0u;
                     const bool promote = !packed_default && e->scan_packed < 0 && rpl == 1 && count >= 48; // (launches the lock-step scan would take)
                     use_split = (e->split_on || probe) && (packed_default || promote);
                     return packed_default || (promote && use_split);
                 }())
        {
            const bool split = use_split;
            if (split)
            {
                // long scans apart (cc_k_scan.h): the packed scan hands points that are still scanning after SCAN_CAP visits to k_scan2_long, which
                // keeps every lane busy with one of them; k_scan2_epi finishes the columns that had such a point
                const dim3 long_grid((unsigned) count, cck::SCAN_LONG_BLOCKS), epi_grid((unsigned) count, cck::SCAN_EPI_BLOCKS);
                if (rpl == 1)
                {
                    hipLaunchKernelGGL((cck::k_scan2<1, false, true>), scan_grid, dim3(64), 0, sc, gs, e->cfg, e->P, e->d_states, first_stream, slot);
                    hipLaunchKernelGGL(cck::k_scan2_long<1>, long_grid, dim3(64), 0, sc, g, e->cfg, e->P, e->d_states, first_stream, slot, e->d_bail_count);
                    hipLaunchKernelGGL(cck::k_scan2_epi<1>, epi_grid, dim3(64), 0, sc, g, e->cfg, e->P, e->d_states, first_stream, slot, e->d_bail_count);
                }
                else
                {
                    hipLaunchKernelGGL((cck::k_scan2<2, false, true>), scan_grid, dim3(64), 0, sc, gs, e->cfg, e->P, e->d_states, first_stream, slot);
                    hipLaunchKernelGGL(cck::k_scan2_long<2>, long_grid, dim3(64), 0, sc, g, e->cfg, e->P, e->d_states, first_stream, slot, e->d_bail_count);
                    hipLaunchKernelGGL(cck::k_scan2_epi<2>, epi_grid, dim3(64), 0, sc, g, e->cfg, e->P, e->d_states, first_stream, slot, e->d_bail_count);
                }
            }
            else if (rpl == 1 && !g.mirror_fields)
                hipLaunchKernelGGL((cck::k_scan2<1, false>), scan_grid, dim3(64), 0, sc, gs, e->cfg, e->P, e->d_states, first_stream, slot);
            else if (rpl == 1)
                hipLaunchKernelGGL((cck::k_scan2<1, true>), scan_grid, dim3(64), 0, sc, gs, e->cfg, e->P, e->d_states, first_stream, slot);
            else if (!g.mirror_fields)
                hipLaunchKernelGGL((cck::k_scan2<2, false>), scan_grid, dim3(64), 0, sc, gs, e->cfg, e->P, e->d_states, first_stream, slot);
            else
                hipLaunchKernelGGL((cck::k_scan2<2, true>), scan_grid, dim3(64), 0, sc, gs, e->cfg, e->P, e->d_states, first_stream, slot);
        }
        else if (rpl == 1 && !g.mirror_fields)
            hipLaunchKernelGGL((cck::k_scan<1, false>), scan_grid, dim3(64), 0, sc, gs, e->cfg, e->P, e->d_states, first_stream, slot);
        else if (rpl == 1)
            hipLaunchKernelGGL((cck::k_scan<1, true>), scan_grid, dim3(64), 0, sc, gs, e->cfg, e->P, e->d_states, first_stream, slot);
        else if (!g.mirror_fields)
            hipLaunchKernelGGL((cck::k_scan<2, false>), scan_grid, dim3(64), 0, sc, gs, e->cfg, e->P, e->d_states, first_stream, slot);
        else
            hipLaunchKernelGGL((cck::k_scan<2, true>), scan_grid, dim3(64), 0, sc, gs, e->cfg, e->P, e->d_states, first_stream, slot);
        CC_MARK(sc); // ev5: scan
        if (sc != sa)
        {
            CC_HIP_CHECK(e, hipEventRecord(e->ev_seg[slot], sc));
            CC_HIP_CHECK(e, hipStreamWaitEvent(sa, e->ev_seg[slot], 0));
        }
        // ---- association + publish chain -------------------------------------------------------------------------
        CC_MARK(sa); // ev6: start of the third chain
        // batch-parallel association in front of the serial kernels: it takes every group of columns in which nothing can differ from the
        // reference's sequential semantics (cc_assocb.h) and stops in front of the first group that might. With k_assoc3 behind it the pair runs
        // assoc_rounds times: a LIMITED launch of the serial kernel takes that one group, the batch-parallel kernel continues behind it; the last
        // serial launch takes whatever is left of the batch.
        auto launch_assocb = [&]()
        {
            if (rpl == 1)
                hipLaunchKernelGGL(cck::k_assocb<1>, dim3(count), dim3(cck::AB_THREADS), 0, sa, g, e->cfg, e->P, e->d_states, first_stream, slot,
                                   e->d_bail_count);
            else
                hipLaunchKernelGGL(cck::k_assocb<2>, dim3(count), dim3(cck::AB_THREADS), 0, sa, g, e->cfg, e->P, e->d_states, first_stream, slot,
                                   e->d_bail_count);
        };
        bool marked7 = false;
        bool global_done = false; // k_associate's work was done inside the last k_assoc3 launch
        // a lean small call (k_small_front in front, results mirrored): k_assocb, then ONE kernel for the serial fall-backs, the ids and the mirror
        const bool small_tail = small_front && e->capture_mirror.state != nullptr && batch_assoc && e->assoc_waves >= 2 && rpl == 1 &&
                                e->cfg.cluster_point_trees_every_nth_column == 1 && !e->debug_no_assoc_fallback;
        if (small_tail)
        {
            launch_assocb();
            CC_MARK(sa); // ev7
            marked7 = true;
            hipLaunchKernelGGL(cck::k_small_tail<1>, dim3(1), dim3(cck::A3_THREADS), 0, sa, g, e->cfg, e->P, e->d_states, first_stream, slot, e->capture_mirror);
            global_done = true;
        }
        // k_assoc3 walks the finished-cluster checks of several columns at once and assumes one check per column
        else if (e->assoc_waves >= 2 && e->cfg.cluster_point_trees_every_nth_column == 1)
        {
            // with or without the links wave (cc_assoc3.h: A3_THREADS): by default (assoc_waves = 0) with it while the streams are few
            // enough for the association chain to be what the step waits for
            const bool lwave = e->assoc_waves == 4 || (e->assoc_waves_auto && count <= CC_LWAVE_MAX_STREAMS);
            const dim3 block(lwave ? cck::A3_THREADS : 192);
            const int rounds = batch_assoc ? (e->assoc_rounds > 0 ? e->assoc_rounds : adaptive_rounds) : 1;
            for (int r = 0; r < rounds; r++)
            {
                if (batch_assoc)
                {
                    launch_assocb();
                    if (r == 0)
                    {
                        CC_MARK(sa); // ev7: the batch-parallel kernel alone ("assoc_lds_ms"); the serial kernels behind it count as "assoc_global_ms"
                        marked7 = true;
                    }
                }
                const int limited = r + 1 < rounds ? 1 : 0;
                // behind k_assocb the serial kernel is a safety net that finds nothing to do: a few blocks sweep over all streams instead of one block
                // per stream waiting for 45 KB of LDS on a busy CU. One block per stream when it is what associates, or while k_assocb has had to stop
                // lately (adaptive_rounds > 1), or when the caller pinned the number of rounds
                const int blocks = (batch_assoc && e->assoc_rounds == 0 && adaptive_rounds == 1 && !e->capturing) ? (count < e->assoc_sweep_blocks ? count : e->assoc_sweep_blocks) : count;
                if (batch_assoc && e->debug_no_assoc_fallback)
                    continue;
                if (rpl == 1)
                    hipLaunchKernelGGL(cck::k_assoc3<1>, dim3(blocks), block, 0, sa, gs, e->cfg, e->P, e->d_states, first_stream, slot, limited, count, 1);
                else
                    hipLaunchKernelGGL(cck::k_assoc3<2>, dim3(blocks), block, 0, sa, gs, e->cfg, e->P, e->d_states, first_stream, slot, limited, count, 1);
                global_done = limited == 0; // (the last launch of k_assoc3 takes the streams that continue in global memory with it)
            }
        }
        else
        {
            if (batch_assoc)
            {
                launch_assocb();
                CC_MARK(sa);
                marked7 = true;
            }
            if (rpl == 1)
                hipLaunchKernelGGL(cck::k_assoc_lds<1>, dim3(count), dim3(64), 0, sa, gs, e->cfg, e->P, e->d_states, first_stream, slot);
            else
                hipLaunchKernelGGL(cck::k_assoc_lds<2>, dim3(count), dim3(64), 0, sa, gs, e->cfg, e->P, e->d_states, first_stream, slot);
        }
        if (!marked7)
            CC_MARK(sa); // ev7: assoc_lds (without the batch-parallel kernel: the serial LDS kernel)
        // streams whose unfinished trees do not fit the LDS pool (or exotic window configs) continue in global memory
        if ((batch_assoc && e->debug_no_assoc_fallback) || global_done)
            ;
        else if (rpl == 1)
            hipLaunchKernelGGL(cck::k_associate<1>, dim3(count), dim3(64), 0, sa, g, e->cfg, e->P, e->d_states, first_stream, slot);
        else
            hipLaunchKernelGGL(cck::k_associate<2>, dim3(count), dim3(64), 0, sa, g, e->cfg, e->P, e->d_states, first_stream, slot);
        CC_MARK(sa); // ev8: assoc_global
        // (without the host synchronisation behind k_insert_par nobody else reads the counter of k_assocb's stops: four bytes ride along here)
        if (batch_assoc && !gate && !gate2 && e->h_bail_count && !e->capturing)
            CC_HIP_CHECK(e, hipMemcpyAsync(e->h_bail_count, e->d_bail_count, 3 * sizeof(int), hipMemcpyDeviceToHost, sa));
        // The ids of the published columns only read what the association of THIS batch left behind (tree root of every cell, cluster id
        // at the root cell; neither is touched again before the ring wraps), so in the pipelined mode they are written on a stream of their
        // own and the next batch's association starts without waiting for them.
        hipStream_t spub = (si != sa && e->publish_off_chain) ? e->stream6 : sa;
        if (spub != sa)
        {
            CC_HIP_CHECK(e, hipEventRecord(e->ev_pubrdy[slot], sa));
            CC_HIP_CHECK(e, hipStreamWaitEvent(spub, e->ev_pubrdy[slot], 0));
        }
        if (!small_tail)
            hipLaunchKernelGGL(cck::k_publish, dim3((unsigned) count, cck::PUBLISH_BLOCKS), dim3(64), 0, spub, g, e->P, e->d_states, first_stream,
                               slot, e->capture_mirror);
        CC_MARK(spub); // ev9: publish
        if (si != sa)
        {
            CC_HIP_CHECK(e, hipEventRecord(e->ev_assoc[slot], spub)); // the batch descriptor slot is free again after its publish
            e->assoc_pending[slot] = true;
            if (rel_seq)
            {
                // every kernel that reads the call's input buffers is ordered in front of this point (insertion -> segmentation -> association -> publish)
                CC_HIP_CHECK(e, hipEventRecord(e->ev_rel[rel_seq % cc_engine::REL_RING], spub));
                e->rel_recorded[rel_seq % cc_engine::REL_RING] = rel_seq;
            }
        }
        CC_HIP_CHECK(e, hipGetLastError());
        if (e->host_prof && hp_gated)
        {
            e->hp_post += std::chrono::duration<double>(std::chrono::steady_clock::now() - hp_t1).count();
            e->hp_calls++;
        }
        return CC_OK;
    };
#undef CC_MARK
    if (lazy_registered)
    {
        const long long my_ntotal = e->cur_ntotal, my_f0 = e->cur_f0;
        const int my_prep_buf = e->prep_buf;
        e->deferred_tail = [=](const std::function<int()>* redo) mutable -> int
        {
            CC_HIP_CHECK(e, hipEventSynchronize(e->ev_gate[slot]));
            const bool fb = gate_h_left[0] != 0, seg = gate_h_left[1] != 0;
            if (!fb && !seg)
            {
                e->lazy_miss = 0;
                return tail(0, 0, 1);
            }
            // some stream's batch was not taken completely (or is not fused): the other insertion kernels / k_table and k_seg_pre, then the chains,
            // then — a call of limit_columns — the continuation passes; all of it with this batch's buffers, not the next one's
            if (++e->lazy_miss >= 2)
                e->lazy_ok = false; // (streams that are not in the steady single-column shape: the plain gate from now on)
            const long long keep_ntotal = e->cur_ntotal, keep_f0 = e->cur_f0;
            const int keep_buf = e->prep_buf;
            e->cur_ntotal = my_ntotal, e->cur_f0 = my_f0, e->prep_buf = my_prep_buf;
            int rc = tail(fb ? 1 : 0, seg ? 1 : 0, 0);
            if (!rc && hipStreamSynchronize(si) != hipSuccess)
                rc = CC_ERR_HIP;
            if (!rc && *e->h_remaining != 0)
                rc = finish_batch(e);
            e->cur_ntotal = keep_ntotal, e->cur_f0 = keep_f0, e->prep_buf = keep_buf;
            if (!rc && redo)
            {
                e->lazy_redone++;
                rc = (*redo)();
            }
            return rc;
        };
        e->lazy_pending = true;
        e->lazy_prev_left = gate_left;
        return CC_OK;
    }
    if (defer)
    {
        e->deferred_tail = [tail](const std::function<int()>*) mutable -> int { return tail(-1, -1, -1); };
        return CC_OK;
    }
    return tail(-1, -1, -1);
}

__global__ void k_clear_remaining(int* remaining)
{
    *remaining = 0;
}

__global__ void k_clear_events(StreamState* states, int first_stream, int count)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count)
    {
        states[first_stream + i].n_events = 0;
        states[first_stream + i].n_links = 0;
    }
}

int collect_events(cc_engine* e, int first_stream, int count);
int collect_links(cc_engine* e, int stream, const StreamState& st);

int resolve_timing(cc_engine* e)
{
    // 10 events per pass: ev0 | prep | ev1 | insert | ev2 ... ev3 | table+segment | ev4 | scan | ev5 ... ev6 | assoc_lds | ev7 |
    // assoc_global | ev8 | publish | ev9
    static const int from[7] = {0, 1, 3, 4, 6, 7, 8};
    for (size_t i = 0; i + 9 < e->ev_used; i += 10)
    {
        for (int k = 0; k < 7; k++)
        {
            float ms = 0.f;
            CC_HIP_CHECK(e, hipEventElapsedTime(&ms, e->ev_pool[i + from[k]], e->ev_pool[i + from[k] + 1]));
            e->kernel_ms[k] += ms;
        }
        e->kernel_launches++;
    }
    for (size_t i = 0; i + 1 < e->pev_used; i += 2)
    {
        float ms = 0.f;
        CC_HIP_CHECK(e, hipEventElapsedTime(&ms, e->pev_pool[i], e->pev_pool[i + 1]));
        e->prep_ahead_ms += ms;
    }
    e->pev_used = 0;
    e->ev_used = 0;
    return CC_OK;
}

int sync_all(cc_engine* e)
{
    {
        int rcf = flush_deferred(e);
        if (rcf)
            return rcf;
    }
    if (e->idle)
        return CC_OK; // (a resident kernel idling on `stream` is not work in flight: it stays)
    {
        int rcs = stop_resident(e);
        if (rcs)
            return rcs;
    }
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream2));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream3));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream4));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream5));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream6));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream7));
    for (bool& b : e->assoc_pending)
        b = false;
    e->idle = true;
    return CC_OK;
}

// ---- lifetime of the caller's input buffers -----------------------------------------------------------------------------------------------
// option "check_input_lifetime": a position-weighted 64-bit sum of the three buffers of a call, taken when the call is made and again when the
// engine reports the buffers as released. A caller that re-uses a buffer too early (a data race the engine cannot see otherwise: the symptom is
// "reset_required" on a healthy stream) changes the sum: the release then fails with CC_ERR_INVALID_ARGUMENT and says which call.
__global__ __launch_bounds__(256) void k_input_sum(const unsigned* __restrict__ w, unsigned long long nwords, const unsigned char* __restrict__ tail, int ntail,
                                                   unsigned long long* __restrict__ out)
{
    unsigned long long acc = 0ull;
    for (unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (unsigned long long) gridDim.x * blockDim.x)
        acc += (unsigned long long) w[i] * (2ull * i + 1ull);
    if (blockIdx.x == 0 && (int) threadIdx.x < ntail)
        acc += (unsigned long long) tail[threadIdx.x] * (2ull * (nwords + threadIdx.x) + 1ull);
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0 && acc)
        atomicAdd(out, acc);
}

static int input_sum(cc_engine* e, const cc_engine::InputRec& r, unsigned long long* sum)
{
    if (!e->h_input_sum)
    {
        CC_HIP_CHECK(e, hipHostMalloc((void**) &e->h_input_sum, 64));
        int rca = alloc_plane(e, &e->d_input_sum, 8);
        if (rca)
            return rca;
    }
    CC_HIP_CHECK(e, hipMemsetAsync(e->d_input_sum, 0, sizeof(unsigned long long), e->stream));
    const struct
    {
        const void* p;
        size_t b;
    } parts[3] = {{r.xyz, r.b_xyz}, {r.inten, r.b_int}, {r.pose, r.b_pose}};
    for (const auto& q : parts)
    {
        const unsigned long long nw = q.b / 4;
        const int blocks = (int) std::min<unsigned long long>(4096ull, (nw + 255ull) / 256ull + 1ull);
        hipLaunchKernelGGL(k_input_sum, dim3(blocks), dim3(256), 0, e->stream, (const unsigned*) q.p, nw, (const unsigned char*) q.p + nw * 4, (int) (q.b % 4), e->d_input_sum);
    }
    CC_HIP_CHECK(e, hipMemcpyAsync(e->h_input_sum, e->d_input_sum, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
    *sum = *e->h_input_sum;
    return CC_OK;
}

// the inputs of every call up to `upto` will not be read again
static int note_released(cc_engine* e, uint64_t upto)
{
    if (upto <= e->released_seq)
        return CC_OK;
    e->released_seq = upto;
    if (e->live_inputs.empty())
        return CC_OK;
    int rc = CC_OK;
    std::vector<cc_engine::InputRec> keep;
    for (const auto& r : e->live_inputs)
    {
        if (r.seq > upto)
        {
            keep.push_back(r);
            continue;
        }
        unsigned long long now = 0ull;
        int rcs = input_sum(e, r, &now);
        if (rcs)
            return rcs;
        if (now != r.sum && rc == CC_OK)
        {
            e->error = "cc_engine_add_firings_device: the input buffers of call " + std::to_string(r.seq) +
                       " were modified before the engine released them (cc_engine_inputs_released / cc_engine_sync): a data race in the caller";
            rc = CC_ERR_INVALID_ARGUMENT;
        }
        if (e->check_input_lifetime >= 2)
        {
            // poison: whoever still reads "its" old data from here (or handed the engine a buffer twice) meets NaN coordinates, 0xFF intensities, NaN poses
            CC_HIP_CHECK(e, hipMemsetAsync(const_cast<void*>(r.xyz), 0xFF, r.b_xyz, e->stream));
            CC_HIP_CHECK(e, hipMemsetAsync(const_cast<void*>(r.inten), 0xFF, r.b_int, e->stream));
            CC_HIP_CHECK(e, hipMemsetAsync(const_cast<void*>(r.pose), 0xFF, r.b_pose, e->stream));
            CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
        }
    }
    e->live_inputs.swap(keep);
    return rc;
}

int finish_batch_inner(cc_engine* e);

// Wait for everything in flight; run continuation passes while some stream stopped early (limit_columns reached).
int finish_batch(cc_engine* e)
{
    int rc = finish_batch_inner(e);
    if (rc)
        return rc;
    // everything enqueued so far is complete and nothing is held back: every call's inputs are released — but for the call that is being submitted
    // right now, if this is a synchronisation inside cc_engine_add_firings_device (the previous batch's continuation passes, a lazy-gate miss)
    return note_released(e, e->call_seq - ((e->in_submit && e->call_seq > 0) ? 1ull : 0ull));
}

int finish_batch_inner(cc_engine* e)
{
    if (!e->batch_open)
        return sync_all(e);
    while (true)
    {
        int rc = sync_all(e);
        if (rc)
            return rc;
        if (e->timing && (rc = resolve_timing(e)))
            return rc;
        if (e->g.record_events && (rc = collect_events(e, e->last_first, e->last_count)))
            return rc;
        if (*e->h_remaining == 0)
            break;
        hipLaunchKernelGGL(k_clear_remaining, dim3(1), dim3(1), 0, e->stream, e->d_remaining);
        const int slot = (int) (e->batch_seq & 3);
        e->batch_seq++;
        rc = launch_batch(e, e->last_first, e->last_count, e->last_n, e->last_xyz, e->last_int, e->last_pose, false, slot, e->stream,
                          e->stream, e->stream);
        if (rc)
            return rc;
    }
    e->batch_open = false;
    return CC_OK;
}

// Firings [f0, f0 + n) of buffers that hold n_total firings per stream (n_total <= 0: the buffers hold exactly n).
int submit(cc_engine* e, int first_stream, int count, int64_t n, const float* d_xyz, const uint8_t* d_int, const double* d_pose,
           bool pipeline, int64_t n_total = 0, int64_t f0 = 0)
{
    e->small_view_ok = false; // (the streams move on: what the last small call mirrored is history)
    {
        int rcs = stop_resident(e);
        if (rcs)
            return rcs;
    }
    const auto hp_e0 = std::chrono::steady_clock::now();
    std::fill(e->state_cached.begin(), e->state_cached.end(), 0);
    if (n_total <= 0)
    {
        n_total = n;
        f0 = 0;
    }
    int rc;
    bool prepared = false;
    const int next_buf = e->prep_buf ^ 1;
    if (pipeline && e->input_on_engine_stream)
    {
        // the caller produced d_xyz / d_int / d_pose with work enqueued on `stream` (e.g. cc_kitti_convert_frames): the preparation
        // chain reads them from its own stream and has to wait for that work (and, with it, for what the engine queued there before)
        CC_HIP_CHECK(e, hipEventRecord(e->ev_input, e->stream));
        CC_HIP_CHECK(e, hipStreamWaitEvent(e->stream5, e->ev_input, 0));
    }
    if (pipeline && e->batch_open && e->pipelined && !(e->parallel_insert && n >= 64))
    {
        // The per-point preparation of this batch depends on nothing the engine holds: it starts now, on its own stream and
        // into the other staging buffer, while the previous batch is still being inserted.
        if (e->timing)
        {
            while (e->pev_pool.size() < e->pev_used + 2)
            {
                hipEvent_t x;
                CC_HIP_CHECK(e, hipEventCreate(&x));
                e->pev_pool.push_back(x);
            }
            CC_HIP_CHECK(e, hipEventRecord(e->pev_pool[e->pev_used], e->stream5));
        }
        if ((rc = launch_prep(e, count, n, d_xyz, d_pose, next_buf, e->stream5, n_total, f0)))
            return rc;
        if (e->timing)
        {
            CC_HIP_CHECK(e, hipEventRecord(e->pev_pool[e->pev_used + 1], e->stream5));
            e->pev_used += 2;
        }
        prepared = true;
    }
    // (the lazy gate: this call's insertion is enqueued BEFORE the previous one's counters are read — launch_batch; a call that cannot do that
    // settles the previous batch first)
    const bool lazy_next = lazy_eligible(e, count, n, pipeline, prepared) && e->batch_open && e->pipelined;
    if (e->lazy_pending && !lazy_next && (rc = flush_deferred(e)))
        return rc;
    if (e->batch_open && !(lazy_next && e->lazy_pending))
    {
        if (pipeline && e->pipelined)
        {
            // the previous batch's insertion chain must be complete before its successor starts; its association chain
            // keeps running on stream2 while this batch is inserted
            CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
            if (*e->h_remaining != 0)
            {
                // continuation passes of the previous batch still read its staged points: this batch's preparation went to the
                // other buffer, so nothing is lost
                if ((rc = finish_batch(e)))
                    return rc;
            }
        }
        else if ((rc = finish_batch(e)))
            return rc;
    }
    e->prep_buf = next_buf;
    e->cur_ntotal = n_total;
    e->cur_f0 = f0;
    const int slot = (int) (e->batch_seq & 3);
    e->batch_seq++;
    hipStream_t si = e->stream, sb = pipeline ? e->stream2 : e->stream, sa = pipeline ? e->stream3 : e->stream;
    hipStream_t sc = (pipeline && e->pipeline_depth >= 2) ? e->stream4 : sb;
    hipStream_t sp = pipeline ? e->stream5 : si;
    if (pipeline && e->assoc_pending[slot])
    {
        // the descriptor slot of batch b - 4 must have been consumed
        CC_HIP_CHECK(e, hipStreamWaitEvent(si, e->ev_assoc[slot], 0));
        e->assoc_pending[slot] = false;
    }
    if (!use_small_front(e, count, n, pipeline)) // (k_small_front begins the batch itself)
        hipLaunchKernelGGL(k_begin_batch, dim3((count + 255) / 256), dim3(256), 0, si, e->d_states, first_stream, count, e->d_remaining,
                           pipeline ? 1 : 0, slot, e->lazy_pending ? e->lazy_prev_left : (const int*) nullptr, pipeline ? e->d_par_left + 2 * slot : (int*) nullptr);
    if (prepared)
    {
        CC_HIP_CHECK(e, hipEventRecord(e->ev_prep[slot], e->stream5));
        CC_HIP_CHECK(e, hipStreamWaitEvent(si, e->ev_prep[slot], 0));
    }
    if (e->host_prof)
        e->hp_entry += std::chrono::duration<double>(std::chrono::steady_clock::now() - hp_e0).count();
    rc = launch_batch(e, first_stream, count, n, d_xyz, d_int, d_pose, true, slot, si, sb, sa, sc, prepared ? si : sp, prepared);
    if (rc)
        return rc;
    e->last_xyz = d_xyz;
    e->last_int = d_int;
    e->last_pose = d_pose;
    e->last_n = n;
    e->last_first = first_stream;
    e->last_count = count;
    e->batch_open = true;
    e->pipelined = pipeline;
    return CC_OK;
}

// Tree links logged by the association kernels during the last call (Geometry::mirror_fields): root cells -> (global column, row)
int collect_links(cc_engine* e, int stream, const StreamState& st)
{
    if (!e->g.mirror_fields || st.n_links <= 0)
        return CC_OK;
    if (st.n_links > e->g.link_capacity)
    {
        e->error = "tree-link log overflow (more than " + std::to_string(e->g.link_capacity) + " links in one call)";
        return CC_ERR_CAPACITY;
    }
    std::vector<int2> raw((size_t) st.n_links);
    CC_HIP_CHECK(e, hipMemcpy(raw.data(), e->P.link_log + (size_t) stream * e->g.link_capacity, raw.size() * sizeof(int2), hipMemcpyDeviceToHost));
    const int R = e->g.num_rows, RC = e->g.ring_cols;
    // a root cell's column is the latest global column <= the last segmented one that maps to its ring slot
    const int64_t last = st.first_unfinished - 1;
    auto gcol_of = [&](int cell)
    {
        const int64_t lc = cell / R;
        int64_t g = last - (((last % RC) - lc + RC) % RC);
        return g;
    };
    auto& dst = e->pending_links[stream];
    for (const int2& l : raw)
    {
        dst.push_back(gcol_of(l.x));
        dst.push_back(l.x % R);
        dst.push_back(gcol_of(l.y));
        dst.push_back(l.y % R);
    }
    return CC_OK;
}

int collect_events(cc_engine* e, int first_stream, int count)
{
    std::vector<StreamState> st(count);
    CC_HIP_CHECK(e, hipMemcpy(st.data(), e->d_states + first_stream, count * sizeof(StreamState), hipMemcpyDeviceToHost));
    for (int i = 0; i < count; i++)
    {
        int n = st[i].n_events;
        if (n <= 0)
            continue;
        auto& dst = e->pending_events[first_stream + i];
        size_t old = dst.size();
        dst.resize(old + n);
        CC_HIP_CHECK(e, hipMemcpy(dst.data() + old, e->P.events + (size_t) (first_stream + i) * e->g.event_capacity,
                                  n * sizeof(cc_event), hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < count; i++)
    {
        int rc = collect_links(e, first_stream + i, st[i]);
        if (rc)
            return rc;
    }
    hipLaunchKernelGGL(k_clear_events, dim3((count + 255) / 256), dim3(256), 0, e->stream, e->d_states, first_stream, count);
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
    return CC_OK;
}

// CC_ERR_RING_OVERRUN: several columns of a batch can be stale and they are segmented in parallel; the reference throws at the lowest
// one, at the first stale cell of its bottom-up row walk (cc.cpp:314-345). The kernel recorded the lowest column; read its cells.
void fixup_overrun(cc_engine* e, int stream, StreamState& st)
{
    if (st.error != CC_ERR_RING_OVERRUN || st.overrun_col == std::numeric_limits<int64_t>::max())
        return;
    const int R = e->g.num_rows;
    std::vector<uint16_t> col((size_t) R);
    const int64_t RC = e->g.ring_cols;
    const size_t off = (size_t) stream * (size_t) e->g.cells + (size_t) (st.overrun_col % RC) * R;
    if (hipMemcpy(col.data(), e->P.gtag + off, (size_t) R * sizeof(uint16_t), hipMemcpyDeviceToHost) != hipSuccess)
        return;
    // cells carry the pass over the ring that filled them (cc_kernels.h: cell_tag); 0 = cleared
    const unsigned tag = 0x8000u | ((unsigned) (st.overrun_col / RC) & 0x7fffu);
    for (int row = R - 1; row >= 0; row--)
        if (col[(size_t) row] != tag && col[(size_t) row] != 0)
        {
            st.error_a = st.overrun_col - (int64_t) ((tag - (unsigned) col[(size_t) row]) & 0x7fffu) * RC; // the stale global column index
            st.error_b = st.overrun_col;
            return;
        }
}

// Text of a kernel-side error; CC_ERR_RING_OVERRUN carries the reference's wording and numbers (cc.cpp:337-342: stale global column
// index found in the cell, column being segmented, ring size).
void set_kernel_error(cc_engine* e, int stream, StreamState& st)
{
    fixup_overrun(e, stream, st);
    char buf[320];
    if (st.error == CC_ERR_RING_OVERRUN)
        snprintf(buf, sizeof(buf), "stream %d: This column is not cleared (ring buffer full or written after clearing): %lld, %lld, %d", stream,
                 (long long) st.error_a, (long long) st.error_b, e->g.ring_cols);
    else
        snprintf(buf, sizeof(buf), "stream %d: kernel-side error %d (%lld, %lld)", stream, st.error, (long long) st.error_a,
                 (long long) st.error_b);
    e->error = buf;
}

int first_stream_error(cc_engine* e, int first_stream, int count)
{
    std::vector<StreamState> st(count);
    if (hipMemcpy(st.data(), e->d_states + first_stream, count * sizeof(StreamState), hipMemcpyDeviceToHost) != hipSuccess)
        return CC_ERR_HIP;
    for (int i = 0; i < count; i++)
        if (st[i].error)
        {
            set_kernel_error(e, first_stream + i, st[i]);
            return st[i].error;
        }
    return CC_OK;
}

constexpr int64_t SMALL_MAX = 8;   // firings per call served by the captured-graph path
constexpr int64_t SMALL_STAGE = 63; // firings the pinned staging holds: calls of up to that many firings can be ONE direct launch of k_small_all
constexpr int SMALL_EVENTS = 64;   // events copied back together with the state

void destroy_small_graphs(cc_engine* e)
{
    for (auto& g : e->small_graphs)
        (void) hipGraphExecDestroy(g.exec);
    e->small_graphs.clear();
}

// One host call = one graph launch + one stream synchronisation: H2D of the packed firings, every kernel of the path, D2H of the
// stream's scalar state and its first events. Returns -1 when the call is not eligible (the caller then takes the general path).
int add_firings_small(cc_engine* e, int stream, int64_t n, const float* xyz, const uint8_t* intensity, const double* poses)
{
    // the whole call as one kernel (k_small_all), launched directly: no graph to capture per call size, any n the small-call kernels take
    const bool direct_ok = e->small_direct && e->small_all && use_small_front(e, 1, n, false) && e->assoc_batch && e->assoc_waves >= 2 &&
                           e->cfg.cluster_point_trees_every_nth_column == 1 && !e->debug_no_assoc_fallback && n <= SMALL_STAGE;
    if (!e->allow_graphs || e->timing || (n > SMALL_MAX && !direct_ok) || e->g.debug_flags)
        return -1;
    const int R = e->g.num_rows;
    const size_t b_xyz = (size_t) SMALL_STAGE * R * 3 * sizeof(float), b_int = (((size_t) SMALL_STAGE * R) + 15) & ~(size_t) 15,
                 b_pose = (size_t) SMALL_STAGE * 12 * sizeof(double);
    if (e->h_small && !e->d_small && alloc_plane(e, &e->d_small, b_xyz + b_int + b_pose) != CC_OK)
        return -1;
    if (!e->h_small)
    {
        if (hipHostMalloc((void**) &e->h_small, b_xyz + b_int + b_pose) != hipSuccess ||
            hipHostMalloc((void**) &e->h_small_seq, 64) != hipSuccess ||
            hipHostMalloc((void**) &e->h_small_state, sizeof(StreamState)) != hipSuccess ||
            hipHostMalloc((void**) &e->h_small_events, SMALL_EVENTS * sizeof(cc_event)) != hipSuccess)
            return -1;
        int rc = alloc_plane(e, &e->d_small, b_xyz + b_int + b_pose);
        if (rc)
            return -1;
    }
    const float* d_xyz = (const float*) e->d_small;
    const uint8_t* d_int = e->d_small + b_xyz;
    const double* d_pose = (const double*) (e->d_small + b_xyz + b_int);
    if (e->ego_capacity < 4096)
    {
        for (int i = 0; i < 4; i++)
            if (alloc_plane(e, &e->d_ego[i], (size_t) 4096 * cck::EGO_STRIDE) != CC_OK)
                return -1;
        e->ego_capacity = 4096;
        e->small_graphs_stale = true;
    }
    if (e->small_graphs_stale)
    {
        destroy_small_graphs(e);
        e->small_graphs_stale = false;
    }
    // With k_small_front the call's inputs are read once, by that kernel, straight from the pinned staging (zero copy), and the results are
    // written to pinned memory by the last kernel (k_publish's mirror): the graph has no copy nodes and no k_begin_batch — five kernels.
    bool lean = use_small_front(e, 1, n, false);
    void *zx = nullptr, *zs = nullptr, *zev = nullptr, *zr = nullptr, *zq = nullptr;
    if (lean && !e->d_small_seq)
    {
        if (alloc_plane(e, &e->d_small_seq, 2) != CC_OK || hipMemset(e->d_small_seq, 0, 16) != hipSuccess)
            lean = false;
        else
        {
            e->h_small_seq[0] = 0;
            e->h_small_seq[1] = 0; // (HostMirror::tail_req)
            e->small_seq_expected = 0;
        }
    }
    if (lean)
        lean = hipHostGetDevicePointer(&zx, e->h_small, 0) == hipSuccess && hipHostGetDevicePointer(&zs, e->h_small_state, 0) == hipSuccess &&
               hipHostGetDevicePointer(&zev, e->h_small_events, 0) == hipSuccess && hipHostGetDevicePointer(&zr, e->h_remaining, 0) == hipSuccess &&
               hipHostGetDevicePointer(&zq, e->h_small_seq, 0) == hipSuccess;
    if (lean)
    {
        d_xyz = (const float*) zx;
        d_int = (const uint8_t*) zx + b_xyz;
        d_pose = (const double*) ((const unsigned char*) zx + b_xyz + b_int);
    }
    // the views of the call's columns ride along with the mirrored results (HostMirror::view): pinned planes of MV_COLS * rows cells + a header
    e->small_view_ok = false;
    long long* zvh = nullptr;
    char* zvb = nullptr;
    if (lean && e->mirror_views && e->g.record_events)
    {
        size_t vbytes = 0;
        (void) cck::view_layout(nullptr, (size_t) cck::MV_COLS * R, &vbytes);
        if (!e->h_small_view_hdr && (hipHostMalloc((void**) &e->h_small_view_hdr, 256) != hipSuccess || hipHostMalloc((void**) &e->h_small_view, vbytes + 64) != hipSuccess))
            return -1;
        void *a = nullptr, *b = nullptr;
        if (hipHostGetDevicePointer(&a, e->h_small_view_hdr, 0) == hipSuccess && hipHostGetDevicePointer(&b, e->h_small_view, 0) == hipSuccess)
            zvh = (long long*) a, zvb = (char*) b;
    }
    auto with_views = [&](cck::HostMirror hm) -> cck::HostMirror
    {
        hm.view_hdr = zvh;
        hm.view = zvh ? zvb : nullptr;
        hm.view_rows = R;
        return hm;
    };
    // option "resident": no launch per call at all — the call is handed to k_resident through a doorbell in pinned memory (cc_k_publish.h)
    const bool resident = lean && direct_ok && e->resident_opt && e->g.num_streams == 1 && xyz != nullptr;
    if (!resident)
    {
        int rcs = stop_resident(e);
        if (rcs)
            return rcs;
    }
    // (a captured one-node graph starts ~3 us sooner than a direct launch — 36.9 against 39.6 us per one-firing call —, so the sizes that have a graph keep it)
    const bool direct = lean && direct_ok && n > SMALL_MAX && !resident;
    if (!direct && !resident && n > SMALL_MAX)
        return -1;
    hipGraphExec_t exec = nullptr;
    if (!resident)
        for (auto& g : e->small_graphs)
            if (g.stream == stream && g.n == n && g.record == e->g.record_events)
                exec = g.exec;
    if (direct || resident)
    {
        // (the resident kernel bakes the staging planes' pointers in: sized for the largest call it takes, once)
        if (ensure_prep(e, (size_t) (resident ? SMALL_STAGE : n) * R) != CC_OK)
            return -1;
    }
    else if (!exec)
    {
        if (ensure_prep(e, (size_t) n * R) != CC_OK)
            return -1;
        e->cur_ntotal = n; // the staged call is a whole buffer of its own
        e->cur_f0 = 0;
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) != hipSuccess)
            return -1;
        bool ok = true;
        if (lean)
            e->capture_mirror = with_views(cck::HostMirror{(StreamState*) zs, (cc_event*) zev, e->g.record_events ? SMALL_EVENTS : 0, (int*) zr, e->d_remaining,
                                                           (unsigned long long*) zq, e->d_small_seq, (unsigned long long*) zq + 1});
        else
        {
            ok = hipMemcpyAsync(e->d_small, e->h_small, b_xyz + b_int + b_pose, hipMemcpyHostToDevice, e->stream) == hipSuccess;
            hipLaunchKernelGGL(k_begin_batch, dim3(1), dim3(64), 0, e->stream, e->d_states, stream, 1, e->d_remaining, 0, 0);
        }
        e->capturing = true;
        ok = ok && launch_batch(e, stream, 1, n, d_xyz, d_int, d_pose, true, 0, e->stream, e->stream, e->stream) == CC_OK;
        e->capturing = false;
        const bool mirrored = e->capture_mirror.state != nullptr;
        e->capture_mirror = cck::HostMirror{nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (!mirrored)
        {
            ok = ok && hipMemcpyAsync(e->h_small_state, e->d_states + stream, sizeof(StreamState), hipMemcpyDeviceToHost, e->stream) == hipSuccess;
            if (e->g.record_events)
                ok = ok && hipMemcpyAsync(e->h_small_events, e->P.events + (size_t) stream * e->g.event_capacity, SMALL_EVENTS * sizeof(cc_event),
                                          hipMemcpyDeviceToHost, e->stream) == hipSuccess;
        }
        if (hipStreamEndCapture(e->stream, &graph) != hipSuccess || !ok || !graph)
        {
            if (graph)
                (void) hipGraphDestroy(graph);
            e->allow_graphs = false; // capture is not possible in this environment: keep using the general path
            return -1;
        }
        const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void) hipGraphDestroy(graph);
        if (ie != hipSuccess)
        {
            e->allow_graphs = false;
            return -1;
        }
        if (e->small_graphs.size() >= 16)
            destroy_small_graphs(e);
        e->small_graphs.push_back({stream, n, e->g.record_events, exec});
    }
    if (!xyz)
        return CC_OK; // (cc_engine_set_option "prewarm_small_graphs": the graph of this call size exists now, nothing is launched)
    memcpy(e->h_small, xyz, (size_t) n * R * 3 * sizeof(float));
    memcpy(e->h_small + b_xyz, intensity, (size_t) n * R);
    memcpy(e->h_small + b_xyz + b_int, poses, (size_t) n * 12 * sizeof(double));
    std::fill(e->state_cached.begin(), e->state_cached.end(), 0);
    const bool was_idle = e->idle;
    auto launch_resident = [&]() -> int
    {
        if (!e->h_res_ctl)
        {
            CC_HIP_CHECK(e, hipHostMalloc((void**) &e->h_res_ctl, sizeof(cck::ResidentCtl)));
            memset(e->h_res_ctl, 0, sizeof(cck::ResidentCtl));
        }
        void* zc = nullptr;
        CC_HIP_CHECK(e, hipHostGetDevicePointer(&zc, e->h_res_ctl, 0));
        __atomic_store_n(&e->h_res_ctl->exited, 0ull, __ATOMIC_RELEASE);
        __atomic_store_n(&e->h_res_ctl->stop, 0ull, __ATOMIC_RELEASE);
        __atomic_store_n(&e->h_res_ctl->calls, 0ull, __ATOMIC_RELEASE);
        const cck::HostMirror hm = with_views(cck::HostMirror{(StreamState*) zs, (cc_event*) zev, e->g.record_events ? SMALL_EVENTS : 0, (int*) zr, e->d_remaining,
                                 (unsigned long long*) zq, e->d_small_seq, (unsigned long long*) zq + 1});
        hipLaunchKernelGGL(cck::k_resident, dim3(1), dim3(cck::AB_THREADS), cck::insert2_lds_bytes(R), e->stream, e->g, e->cfg, planes_with_prep(e, e->prep_buf),
                           e->d_states, stream, 0, d_xyz, d_int, d_pose, e->d_remaining, e->d_ego[0], e->d_bail_count, hm, (cck::ResidentCtl*) zc,
                           (unsigned long long) e->res_idle_ms * 100000ull);
        CC_HIP_CHECK(e, hipGetLastError());
        e->res_running = true;
        e->res_launches++;
        return CC_OK;
    };
    // the kernel has left by itself (watchdog, or the call before needed the host): the stream is free again once it has retired
    auto reap_resident = [&]() -> int
    {
        if (e->res_running && __atomic_load_n(&e->h_res_ctl->exited, __ATOMIC_ACQUIRE) != 0ull)
        {
            CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
            e->res_calls += __atomic_load_n(&e->h_res_ctl->calls, __ATOMIC_ACQUIRE);
            e->res_running = false;
        }
        return CC_OK;
    };
    if (resident)
    {
        int rcr = reap_resident();
        if (rcr)
            return rcr;
        if (!e->res_running)
        {
            if ((rcr = flush_deferred(e)) != CC_OK || (rcr = launch_resident()) != CC_OK)
                return rcr;
        }
        // the doorbell: the firings are in the pinned staging (the memcpys above), the call's number and size go out last
        __atomic_store_n(&e->h_res_ctl->bell, ((e->small_seq_expected + 1ull) << 8) | (unsigned long long) n, __ATOMIC_RELEASE);
    }
    else if (direct)
    {
        int rcf = flush_deferred(e);
        if (rcf)
            return rcf;
        const cck::HostMirror hm = with_views(cck::HostMirror{(StreamState*) zs, (cc_event*) zev, e->g.record_events ? SMALL_EVENTS : 0, (int*) zr, e->d_remaining,
                                 (unsigned long long*) zq, e->d_small_seq, (unsigned long long*) zq + 1});
        hipLaunchKernelGGL(cck::k_small_all, dim3(1), dim3(cck::AB_THREADS), cck::insert2_lds_bytes(R), e->stream, e->g, e->cfg, planes_with_prep(e, e->prep_buf),
                           e->d_states, stream, 0, d_xyz, d_int, d_pose, (long long) n, e->d_remaining, e->d_ego[0], e->d_bail_count, hm);
        CC_HIP_CHECK(e, hipGetLastError());
    }
    else
        CC_HIP_CHECK(e, hipGraphLaunch(exec, e->stream));
    if (lean)
    {
        // the call's last kernel writes the results into pinned memory and then this counter: spinning on it returns as soon as they are
        // there (a stream synchronisation adds the driver's wake-up to every column); the stream itself is in order for whatever follows
        const unsigned long long want = ++e->small_seq_expected;
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false, tail_launched = false;
        // k_small_all left columns to the serial kernels (a stop of the batch-parallel association, a stream that continues in global memory): they,
        // the cluster ids and the mirror are k_small_tail's, launched behind it on the same stream
        auto tail_if_asked = [&]()
        {
            if (tail_launched || __atomic_load_n(e->h_small_seq + 1, __ATOMIC_ACQUIRE) != want)
                return;
            tail_launched = true;
            const cck::HostMirror hm = with_views(cck::HostMirror{(StreamState*) zs, (cc_event*) zev, e->g.record_events ? SMALL_EVENTS : 0, (int*) zr, e->d_remaining,
                                     (unsigned long long*) zq, e->d_small_seq, (unsigned long long*) zq + 1});
            hipLaunchKernelGGL(cck::k_small_tail<1>, dim3(1), dim3(cck::A3_THREADS), 0, e->stream, e->g, e->cfg, e->P, e->d_states, stream, 0, hm);
            e->small_tail_launches++;
        };
        for (unsigned spins = 0;; spins++)
        {
            if (__atomic_load_n(e->h_small_seq, __ATOMIC_ACQUIRE) >= want)
            {
                seen = true;
                break;
            }
            tail_if_asked();
            if (resident && (spins & 31u) == 31u && !tail_launched && __atomic_load_n(&e->h_res_ctl->exited, __ATOMIC_ACQUIRE) != 0ull &&
                __atomic_load_n(e->h_small_seq, __ATOMIC_ACQUIRE) < want && __atomic_load_n(e->h_small_seq + 1, __ATOMIC_ACQUIRE) != want)
            {
                // the kernel left without taking this call (its watchdog fired as the bell rang): launch it again, the bell still stands
                int rcr = reap_resident();
                if (!rcr)
                    rcr = launch_resident();
                if (rcr)
                    return rcr;
            }
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(resident ? 2000 : 20))
                break; // (a first launch loading code, a stop in a debugger: let the driver wait)
        }
        if (!seen && resident)
        {
            // (a stream synchronisation would wait for a kernel that waits for us)
            (void) stop_resident(e);
            if (__atomic_load_n(e->h_small_seq, __ATOMIC_ACQUIRE) < want)
            {
                e->error = "resident kernel: a call was not answered within 2 s";
                return CC_ERR_HIP;
            }
        }
        else if (!seen)
        {
            CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
            tail_if_asked();
            if (tail_launched)
                CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
            if (__atomic_load_n(e->h_small_seq, __ATOMIC_ACQUIRE) < want)
            {
                e->error = "small call: the results were not mirrored";
                return CC_ERR_HIP;
            }
        }
    }
    else
        CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
    e->idle = was_idle; // (capturing the graph went through launch_batch; replaying it only touches `stream`, which is drained again)
    if (resident)
    {
        // a call that needed the host (serial fall-backs, continuation passes) or left an error made the kernel leave: reap it now, so that
        // whatever follows finds the stream free
        int rcr = (*e->h_remaining != 0 || e->h_small_state->error != 0) ? stop_resident(e) : reap_resident();
        if (rcr)
            return rcr;
    }
    if (*e->h_remaining != 0)
    {
        // the kernels stopped early (limit_columns): continue on the general path, which also collects the events
        e->last_xyz = d_xyz;
        e->last_int = d_int;
        e->last_pose = d_pose;
        e->last_n = n;
        e->cur_ntotal = n;
        e->cur_f0 = 0;
        e->last_first = stream;
        e->last_count = 1;
        e->batch_open = true;
        e->pipelined = false;
        int rc = finish_batch(e);
        return rc ? rc : first_stream_error(e, stream, 1);
    }
    StreamState& st = *e->h_small_state;
    if (e->g.record_events && st.n_events > 0)
    {
        auto& dst = e->pending_events[stream];
        const int first = st.n_events < SMALL_EVENTS ? st.n_events : SMALL_EVENTS;
        dst.insert(dst.end(), e->h_small_events, e->h_small_events + first);
        if (st.n_events > SMALL_EVENTS)
        {
            const size_t old = dst.size();
            dst.resize(old + (size_t) (st.n_events - SMALL_EVENTS));
            CC_HIP_CHECK(e, hipMemcpy(dst.data() + old, e->P.events + (size_t) stream * e->g.event_capacity + SMALL_EVENTS,
                                      (size_t) (st.n_events - SMALL_EVENTS) * sizeof(cc_event), hipMemcpyDeviceToHost));
        }
    }
    if (e->g.record_events)
    {
        int rcl = collect_links(e, stream, st);
        if (rcl)
            return rcl;
    }
    if (st.error)
    {
        set_kernel_error(e, stream, st);
        return st.error;
    }
    e->state_cache[stream] = st; // (n_events / n_links are not part of what cc_engine_stream_state reports)
    e->state_cached[stream] = 1;
    // the views of the columns this call's events name came with the results (unless the serial fall-backs finished the call, or they were too many)
    // (the header's column list is there with the results; the views themselves follow within microseconds, stamped with the call's number)
    e->small_view_ok = zvh != nullptr && e->h_small_view_hdr[1] > 0;
    e->small_view_stream = stream;
    return CC_OK;
}

// ---- the engine's HIP streams are a process-wide resource ---------------------------------------------------------------------------------
// A HIP stream gets its hardware queue when it is first used, and queues are handed to the compute pipes in creation order. The four chains of
// the pipelined path only overlap as designed when their queues sit on different pipes: the first engine of a process got that by accident
// (four queues created back to back), every later engine got recycled queues in another order — insertion and association on one pipe — and
// ran 10 - 15 % slower at 32 streams, 6 - 10 % at 256 (tools/step_probe.py). So a set of seven streams is created once per
// device, its queues are materialised in a fixed order (one empty launch each: the four chains first), and the set is kept for the next engine
// when an engine is destroyed. Two engines alive at the same time get two sets.
__global__ void k_touch_stream() {}

struct StreamSet
{
    hipStream_t s[7]{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // stream, stream2 .. stream7
};
std::mutex g_stream_sets_mu;
std::vector<std::pair<int, StreamSet>> g_stream_sets; // (device, set) not in use by a live engine

bool acquire_stream_set(int device, StreamSet* out)
{
    {
        std::lock_guard<std::mutex> lock(g_stream_sets_mu);
        for (size_t i = 0; i < g_stream_sets.size(); i++)
            if (g_stream_sets[i].first == device)
            {
                *out = g_stream_sets[i].second;
                g_stream_sets.erase(g_stream_sets.begin() + (long) i);
                return true;
            }
    }
    int prio_lo = 0, prio_hi = 0;
    (void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi); // numerically lower = higher priority
    // The serial chains (insertion, association) bound a pipelined step; the table / segmentation / scan chain between them is
    // throughput work with slack, so it gets the low-priority queue and yields issue slots and memory bandwidth to the other two.
    const int prio[7] = {prio_hi, prio_lo, prio_hi, prio_lo, prio_lo, prio_lo, prio_hi};
    StreamSet set;
    for (int i = 0; i < 7; i++)
        if (hipStreamCreateWithPriority(&set.s[i], hipStreamNonBlocking, prio[i]) != hipSuccess)
        {
            for (int k = 0; k < i; k++)
                (void) hipStreamDestroy(set.s[k]);
            return false;
        }
    // queues in this order: insertion, segmentation, association, window scan (the four chains), then publish, table, preparation
    for (int i : {0, 1, 2, 3, 5, 6, 4})
    {
        hipLaunchKernelGGL(k_touch_stream, dim3(1), dim3(1), 0, set.s[i]);
        (void) hipStreamSynchronize(set.s[i]);
    }
    *out = set;
    return true;
}

void release_stream_set(int device, const StreamSet& set)
{
    std::lock_guard<std::mutex> lock(g_stream_sets_mu);
    g_stream_sets.push_back({device, set});
}

// (all streams idle) the pooled set goes back for the next engine of the process; CU-masked experiment streams are destroyed
void give_back_streams(cc_engine* e)
{
    hipStream_t all[7] = {e->stream, e->stream2, e->stream3, e->stream4, e->stream5, e->stream6, e->stream7};
    if (e->streams_pooled)
    {
        StreamSet set;
        for (int i = 0; i < 7; i++)
            set.s[i] = all[i];
        release_stream_set(e->device, set);
    }
    else
        for (hipStream_t st : all)
            if (st)
                (void) hipStreamDestroy(st);
    e->stream = e->stream2 = e->stream3 = e->stream4 = e->stream5 = e->stream6 = e->stream7 = nullptr;
}
} // namespace

extern "C" {

void cc_config_default(cc_config* c)
{
    memset(c, 0, sizeof(*c));
    c->is_single_threaded = 0;
    c->sensor_is_clockwise = 1;
    c->num_columns = 1700;
    c->supplement_inclination_angle_for_nan_cells = 1;
    c->max_slope = 0.2f;
    c->first_ring_as_ground_max_allowed_z_diff = 0.4f;
    c->first_ring_as_ground_min_allowed_z_diff = -0.4f;
    c->last_ground_point_slope_higher_than = -0.1f;
    c->last_ground_point_distance_smaller_than = 5.f;
    c->ground_because_close_to_last_certain_ground_max_z_diff = 0.4f;
    c->ground_because_close_to_last_certain_ground_max_dist_diff = 2.0f;
    c->obstacle_because_next_certain_obstacle_max_dist_diff = 0.3f;
    c->use_terrain = 0;
    c->terrain_max_allowed_z_diff = 0.4f;
    c->fog_filtering_enabled = 0;
    c->fog_filtering_intensity_below = 2;
    c->fog_filtering_distance_below = 18.f;
    c->fog_filtering_inclination_above = -0.06f;
    c->max_distance = 0.7f;
    c->max_steps_in_row = 20;
    c->max_steps_in_column = 20;
    c->stop_after_association_enabled = 1;
    c->stop_after_association_min_steps = 1;
    c->ignore_points_in_chessboard_pattern = 1;
    c->ignore_points_with_too_big_inclination_angle_diff = 1;
    c->use_last_point_for_cluster_stamp = 0;
    c->cluster_point_trees_every_nth_column = 1;
}

void cc_config_kitti(cc_config* c)
{
    cc_config_default(c);
    c->is_single_threaded = 1;
    c->num_columns = 2200;
    c->ignore_points_in_chessboard_pattern = 0;
    c->max_distance = 0.5f;
    c->height_ref_to_maximum_ = 0.5f;
    c->height_ref_to_ground_ = -1.7f;
    c->length_ref_to_front_end_ = 3.f;
    c->length_ref_to_rear_end_ = -3.f;
    c->width_ref_to_left_mirror_ = 1.5f;
    c->width_ref_to_right_mirror_ = -1.5f;
}

const char* cc_version(void)
{
    return "continuous_clustering_amd 0.1 gfx950";
}

int cc_engine_create(cc_engine** out, int device, int num_streams, int num_rows, const cc_config* cfg)
{
    if (!out)
        return CC_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int rc = validate(nullptr, cfg, num_rows, num_streams);
    if (rc)
        return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return CC_ERR_NO_DEVICE;
    cc_engine* e = new cc_engine();
    e->device = device;
    (void) hipSetDevice(device);
    // experiment switch CC_OPT_CU_SPLIT="ins,seg,scan,assoc": the four chains of the pipelined path on disjoint sets of compute units (CU-masked
    // streams; the mask's bits are dealt round-robin over XCDs and shader engines by the driver, so a contiguous range is spread over the chip)
    bool cu_split = false;
    if (const char* cs = std::getenv("CC_ENABLE_ENV_OPTS") ? std::getenv("CC_OPT_CU_SPLIT") : nullptr)
    {
        int part[4] = {0, 0, 0, 0};
        hipDeviceProp_t prop;
        if (sscanf(cs, "%d,%d,%d,%d", &part[0], &part[1], &part[2], &part[3]) == 4 && hipGetDeviceProperties(&prop, device) == hipSuccess)
        {
            const int ncu = prop.multiProcessorCount;
            const int words = (ncu + 31) / 32;
            auto masked = [&](hipStream_t* out_stream, int from, int cnt) -> bool
            {
                std::vector<uint32_t> m(words, 0u);
                for (int i = from; i < from + cnt && i < ncu; i++)
                    m[i >> 5] |= 1u << (i & 31);
                return hipExtStreamCreateWithCUMask(out_stream, (uint32_t) words, m.data()) == hipSuccess;
            };
            const int o1 = part[0], o2 = o1 + part[1], o3 = o2 + part[2];
            cu_split = part[0] > 0 && part[1] > 0 && part[2] > 0 && part[3] > 0 && o3 + part[3] <= ncu && masked(&e->stream, 0, part[0]) &&
                       masked(&e->stream5, 0, part[0]) && masked(&e->stream7, 0, part[0]) && masked(&e->stream2, o1, part[1]) &&
                       masked(&e->stream4, o2, part[2]) && masked(&e->stream3, o3, part[3]) && masked(&e->stream6, o3, part[3]);
            if (!cu_split)
            {
                fprintf(stderr, "cc_engine_create: CC_OPT_CU_SPLIT=%s rejected (%d compute units)\n", cs, ncu);
                give_back_streams(e); // (whatever masked streams were created)
                delete e;
                return CC_ERR_INVALID_ARGUMENT;
            }
        }
    }
    if (cu_split)
        fprintf(stderr, "cc_engine_create: chains on disjoint compute units (CC_OPT_CU_SPLIT=%s)\n", std::getenv("CC_OPT_CU_SPLIT"));
    else
    {
        StreamSet set;
        if (hipSetDevice(device) != hipSuccess || !acquire_stream_set(device, &set))
        {
            delete e;
            return CC_ERR_HIP;
        }
        e->stream = set.s[0], e->stream2 = set.s[1], e->stream3 = set.s[2], e->stream4 = set.s[3], e->stream5 = set.s[4], e->stream6 = set.s[5], e->stream7 = set.s[6];
        e->streams_pooled = true;
    }
    for (int i = 0; i < 4; i++)
    {
        (void) hipEventCreateWithFlags(&e->ev_ins[i], hipEventDisableTiming);
        (void) hipEventCreateWithFlags(&e->ev_gate[i], hipEventDisableTiming);
        (void) hipEventCreateWithFlags(&e->ev_seg[i], hipEventDisableTiming);
        (void) hipEventCreateWithFlags(&e->ev_assoc[i], hipEventDisableTiming);
        (void) hipEventCreateWithFlags(&e->ev_segscan[i], hipEventDisableTiming);
        (void) hipEventCreateWithFlags(&e->ev_prep[i], hipEventDisableTiming);
        (void) hipEventCreateWithFlags(&e->ev_pubrdy[i], hipEventDisableTiming);
        (void) hipEventCreateWithFlags(&e->ev_rel[i], hipEventDisableTiming);
        (void) hipEventCreateWithFlags(&e->ev_ego[i], hipEventDisableTiming);
        (void) hipEventCreateWithFlags(&e->ev_rel[i + 4], hipEventDisableTiming);
        if (i == 0)
            (void) hipEventCreateWithFlags(&e->ev_input, hipEventDisableTiming);

    }
    e->cfg = *cfg;
    e->g.num_streams = num_streams;
    e->g.tree_capacity = 1 << 15;
    e->g.record_events = num_streams == 1 ? 1 : 0;
    e->g.mirror_fields = e->g.record_events; // the host mirror's extra per-point fields: on where a host reads columns back (1 stream)
    // tree links logged per stream and call for a host mirror (8 bytes each): with the early stop of the window scan switched off a point links
    // up to LINK_SLOTS trees, and a call may carry several rotations — roomy where a mirror is likely (few streams), small otherwise
    e->g.link_capacity = num_streams <= 8 ? (1 << 20) : 8192;
    fill_geometry(e, num_rows);
    e->g.event_capacity = e->g.record_events ? 3 * (e->g.limit_columns + e->g.num_columns) + 4096 : 1;
    e->pending_events.resize(num_streams);
    e->pending_links.resize(num_streams);
    e->state_cache.resize(num_streams);
    e->state_cached.assign(num_streams, 0);
    (void) hipFuncSetAttribute((const void*) cck::k_seg_scan, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void) hipFuncSetAttribute((const void*) cck::k_insert2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void) hipFuncSetAttribute((const void*) cck::k_small_front, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void) hipFuncSetAttribute((const void*) cck::k_small_all, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void) hipFuncSetAttribute((const void*) cck::k_insert2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void) hipFuncSetAttribute((const void*) (cck::k_insert_par<1, cck::IP_WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    (void) hipFuncSetAttribute((const void*) (cck::k_insert_par<1, 2 * cck::IP_WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            e->num_cus = prop.multiProcessorCount;
    }
    rc = allocate(e);
    if (rc == CC_OK && hipHostMalloc((void**) &e->h_par_left, 8 * sizeof(int)) != hipSuccess)
        rc = CC_ERR_HIP;
    {
        const char* hp = std::getenv("CC_HOST_PROF");
        e->host_prof = hp && hp[0] == '1';
    }
    if (rc == CC_OK && hipHostMalloc((void**) &e->h_bail_count, 4 * sizeof(int)) != hipSuccess)
        rc = CC_ERR_HIP;
    if (rc == CC_OK)
        e->h_bail_count[0] = e->h_bail_count[1] = e->h_bail_count[2] = e->h_bail_count[3] = 0;
    if (rc == CC_OK && hipHostMalloc((void**) &e->h_remaining, sizeof(int)) != hipSuccess)
        rc = CC_ERR_HIP;
    if (rc == CC_OK)
    {
        *e->h_remaining = 0;
        rc = reset_state(e, false);
    }
    if (rc != CC_OK)
    {
        fprintf(stderr, "cc_engine_create: %s\n", e->error.c_str());
        free_all(e);
        if (e->h_remaining)
            (void) hipHostFree(e->h_remaining);
        if (e->h_par_left)
            (void) hipHostFree(e->h_par_left);
        if (e->h_bail_count)
            (void) hipHostFree(e->h_bail_count);
        give_back_streams(e);
        delete e;
        return rc;
    }
    // experiment switches for harnesses that only see the reference's class API (tests/cpp/dropin_demo): CC_OPT_<OPTION>=<value> in the environment
    // (only where CC_ENABLE_ENV_OPTS is set: a library does not change its behaviour on stray environment variables)
    for (const char* name : {"small_front", "seg_small_max", "fuse_front", "small_graphs"})
    {
        if (!getenv("CC_ENABLE_ENV_OPTS"))
            break;
        std::string key = std::string("CC_OPT_") + name;
        for (auto& ch : key)
            ch = (char) toupper((unsigned char) ch);
        if (const char* v = getenv(key.c_str()))
            (void) cc_engine_set_option(e, name, atoll(v));
    }
    *out = e;
    return CC_OK;
}

void cc_engine_destroy(cc_engine* e)
{
    if (!e)
        return;
    (void) hipSetDevice(e->device);
    (void) stop_resident(e);
    if (e->host_prof && e->hp_calls > 0)
        fprintf(stderr, "[cc host_prof] gated calls %lld: entry->launch_batch %.1f us, launch_batch->gate %.1f us, gate wait %.1f us, gate->return %.1f us (per call)\n",
                e->hp_calls, e->hp_entry / e->hp_calls * 1e6, e->hp_pre / e->hp_calls * 1e6, e->hp_gate / e->hp_calls * 1e6, e->hp_post / e->hp_calls * 1e6);
    (void) hipStreamSynchronize(e->stream);
    (void) hipStreamSynchronize(e->stream2);
    (void) hipStreamSynchronize(e->stream3);
    (void) hipStreamSynchronize(e->stream4);
    (void) hipStreamSynchronize(e->stream6);
    (void) hipStreamSynchronize(e->stream7);
    (void) hipStreamSynchronize(e->stream5);
    e->deferred_tail = nullptr; // (chains a pipelined call left to "the next call": there is none)
    e->lazy_pending = false;
    destroy_small_graphs(e);
    free_all(e); // also the pinned small-call staging
    if (e->h_view)
        (void) hipHostFree(e->h_view);
    for (int i = 0; i < 4; i++)
    {
        (void) hipEventDestroy(e->ev_ins[i]);
        (void) hipEventDestroy(e->ev_gate[i]);
        (void) hipEventDestroy(e->ev_seg[i]);
        (void) hipEventDestroy(e->ev_assoc[i]);
        (void) hipEventDestroy(e->ev_segscan[i]);
        (void) hipEventDestroy(e->ev_prep[i]);
        (void) hipEventDestroy(e->ev_pubrdy[i]);
        (void) hipEventDestroy(e->ev_rel[i]);
        (void) hipEventDestroy(e->ev_ego[i]);
        (void) hipEventDestroy(e->ev_rel[i + 4]);
        if (i == 0)
            (void) hipEventDestroy(e->ev_input);
    }
    for (hipEvent_t ev : e->pev_pool)
        (void) hipEventDestroy(ev);
    for (hipEvent_t ev : e->ev_pool)
        (void) hipEventDestroy(ev);
    if (e->h_remaining)
        (void) hipHostFree(e->h_remaining);
    if (e->h_par_left)
        (void) hipHostFree(e->h_par_left);
    if (e->h_bail_count)
        (void) hipHostFree(e->h_bail_count);
    give_back_streams(e);
    delete e;
}

int cc_engine_set_config(cc_engine* e, const cc_config* cfg)
{
    if (!e || !cfg)
        return CC_ERR_INVALID_ARGUMENT;
    destroy_small_graphs(e); // captured graphs bake configuration, geometry and plane pointers
    int rc = validate(e, cfg, e->g.num_rows, e->g.num_streams);
    if (rc)
        return rc;
    (void) hipSetDevice(e->device);
    if ((rc = stop_resident(e)) != CC_OK) // (so does the resident kernel)
        return rc;
    rc = finish_batch(e);
    if (rc)
        return rc;
    (void) hipSetDevice(e->device);
    const bool need_reset = (e->cfg.is_single_threaded != 0) != (cfg->is_single_threaded != 0) ||
                            (e->cfg.sensor_is_clockwise != 0) != (cfg->sensor_is_clockwise != 0) ||
                            e->cfg.num_columns != cfg->num_columns; // cc.cpp:69-74
    e->cfg = *cfg;
    e->g.max_distance_squared = cfg->max_distance * cfg->max_distance; // cc.cpp:80
    const bool force_global = cfg->max_steps_in_row > WIN_COLS - 2;
    if (need_reset || force_global)
    {
        // raise reset_required on every stream; geometry keeps the old num_columns until cc_engine_reset (cc.cpp:14)
        std::vector<StreamState> st(e->g.num_streams);
        CC_HIP_CHECK(e, hipMemcpy(st.data(), e->d_states, st.size() * sizeof(StreamState), hipMemcpyDeviceToHost));
        for (auto& s : st)
        {
            if (need_reset)
                s.reset_required = 1;
            if (force_global)
                s.assoc_mode = 1;
        }
        CC_HIP_CHECK(e, hipMemcpy(e->d_states, st.data(), st.size() * sizeof(StreamState), hipMemcpyHostToDevice));
        std::fill(e->state_cached.begin(), e->state_cached.end(), 0);
    }
    return CC_OK;
}

int cc_engine_reset(cc_engine* e, int num_rows)
{
    if (!e)
        return CC_ERR_INVALID_ARGUMENT;
    destroy_small_graphs(e);
    int rc = validate(e, &e->cfg, num_rows, e->g.num_streams);
    if (rc)
        return rc;
    (void) hipSetDevice(e->device);
    if ((rc = stop_resident(e)) != CC_OK)
        return rc;
    // A pipelined call may have left the chains behind its insertion to the next call (deferred tail / lazy gate): that closure holds the OLD
    // geometry, plane pointers and batch slot by value. Launch it now, against the state it belongs to, so that nothing of the old epoch
    // survives the reset (it would run on freed planes after a change of shape and mark a slot of the new epoch as pending).
    rc = flush_deferred(e);
    if (rc)
        return rc;
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream2));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream3));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream4));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream5));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream6));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream7));
    e->batch_open = false;
    e->idle = true;
    rc = note_released(e, e->call_seq); // (nothing of the old epoch reads the callers' buffers any more)
    if (rc)
        return rc;
    const bool same_shape = num_rows == e->g.num_rows && e->cfg.num_columns == e->g.num_columns;
    if (!same_shape)
    {
        free_all(e);
        fill_geometry(e, num_rows);
        e->g.event_capacity = e->g.record_events ? 3 * (e->g.limit_columns + e->g.num_columns) + 4096 : 1;
        rc = allocate(e);
        if (rc)
            return rc;
    }
    else
        fill_geometry(e, num_rows);
    return reset_state(e, same_shape);
}

int cc_engine_set_robot_from_sensor(cc_engine* e, int stream, const double tf[12])
{
    if (!e || !tf || stream < -1 || stream >= e->g.num_streams)
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    int rc = stop_resident(e); // (the host writes the streams' state below)
    if (rc)
        return rc;
    rc = finish_batch(e);
    if (rc)
        return rc;
    const int first = stream < 0 ? 0 : stream, count = stream < 0 ? e->g.num_streams : 1;
    std::vector<StreamState> st(count);
    CC_HIP_CHECK(e, hipMemcpy(st.data(), e->d_states + first, count * sizeof(StreamState), hipMemcpyDeviceToHost));
    for (auto& s : st)
    {
        memcpy(s.robot_from_sensor, tf, 12 * sizeof(double));
        s.has_robot_tf = 1;
    }
    CC_HIP_CHECK(e, hipMemcpy(e->d_states + first, st.data(), count * sizeof(StreamState), hipMemcpyHostToDevice));
    std::fill(e->state_cached.begin(), e->state_cached.end(), 0);
    return CC_OK;
}

int cc_engine_add_firings(cc_engine* e, int stream, int64_t n, const float* xyz, const uint8_t* intensity, const double* poses)
{
    if (!e || stream < 0 || stream >= e->g.num_streams || n < 0 || (n > 0 && (!xyz || !intensity || !poses)))
        return CC_ERR_INVALID_ARGUMENT;
    if (n == 0)
        return CC_OK;
    (void) hipSetDevice(e->device);
    int rc = finish_batch(e);
    if (rc)
        return rc;
    rc = add_firings_small(e, stream, n, xyz, intensity, poses);
    if (rc >= 0)
        return rc;
    const int R = e->g.num_rows;
    // bound a host batch so that staging stays small and a batch never outruns the clearing front
    const int64_t max_chunk = std::max<int64_t>(1, 2 * (int64_t) e->g.num_columns);
    if (e->stage_capacity < std::min(n, max_chunk))
    {
        int64_t cap = std::min(n, max_chunk);
        if (e->d_stage_xyz)
        {
            // keep the old blocks in `allocations`; they are released with the engine
        }
        if ((rc = alloc_plane(e, &e->d_stage_xyz, (size_t) cap * R * 3)) != 0 || (rc = alloc_plane(e, &e->d_stage_int, (size_t) cap * R)) != 0 ||
            (rc = alloc_plane(e, &e->d_stage_pose, (size_t) cap * 12)) != 0)
            return rc;
        e->stage_capacity = cap;
    }
    for (int64_t off = 0; off < n; off += max_chunk)
    {
        const int64_t m = std::min(max_chunk, n - off);
        CC_HIP_CHECK(e, hipMemcpyAsync(e->d_stage_xyz, xyz + (size_t) off * R * 3, (size_t) m * R * 3 * sizeof(float),
                                       hipMemcpyHostToDevice, e->stream));
        CC_HIP_CHECK(e, hipMemcpyAsync(e->d_stage_int, intensity + (size_t) off * R, (size_t) m * R, hipMemcpyHostToDevice, e->stream));
        CC_HIP_CHECK(e, hipMemcpyAsync(e->d_stage_pose, poses + (size_t) off * 12, (size_t) m * 12 * sizeof(double),
                                       hipMemcpyHostToDevice, e->stream));
        rc = submit(e, stream, 1, m, e->d_stage_xyz, e->d_stage_int, e->d_stage_pose, false);
        if (rc)
            return rc;
        rc = finish_batch(e);
        if (rc)
            return rc;
        rc = first_stream_error(e, stream, 1);
        if (rc)
            return rc;
    }
    return CC_OK;
}

int cc_engine_add_firings_device(cc_engine* e, int64_t n, const float* d_xyz, const uint8_t* d_intensity, const double* d_poses)
{
    if (!e || n < 0 || (n > 0 && (!d_xyz || !d_intensity || !d_poses)))
        return CC_ERR_INVALID_ARGUMENT;
    if (n == 0)
        return CC_OK;
    (void) hipSetDevice(e->device);
    // throughput path: when nobody reads events or columns between batches, batch b + 1 is inserted while batch b is still
    // being segmented and associated (three chains of HIP streams). A large call is cut into sub-batches that enter the
    // pipeline one after the other: same work, but the last firing of the call leaves the pipeline a sub-batch (not a whole
    // call) after it entered.
    const bool pipeline = e->g.record_events == 0 && e->allow_pipeline;
    // the call's number (cc_engine_inputs_released); with "check_input_lifetime" also what its buffers hold right now
    e->call_seq++;
    if (e->check_input_lifetime)
    {
        const size_t cells = (size_t) e->g.num_streams * (size_t) n * (size_t) e->g.num_rows;
        cc_engine::InputRec r{e->call_seq, d_xyz, d_intensity, d_poses, cells * 3 * sizeof(float), cells, (size_t) e->g.num_streams * (size_t) n * 12 * sizeof(double), 0ull};
        int rcs = input_sum(e, r, &r.sum);
        if (rcs)
            return rcs;
        e->live_inputs.push_back(r);
    }
    struct InSubmit
    {
        cc_engine* e;
        explicit InSubmit(cc_engine* e_) : e(e_) { e->in_submit = true; }
        ~InSubmit() { e->in_submit = false, e->cur_call_last = true; }
    } guard(e);
    // (measured: per-launch fixed costs outweigh the shorter fill / drain at 2200-firing calls, so it is off unless asked for)
    int64_t sub = e->sub_batch > 0 ? e->sub_batch : n;
    if (!pipeline || sub >= n)
        return submit(e, 0, e->g.num_streams, n, d_xyz, d_intensity, d_poses, pipeline);
    for (int64_t f0 = 0; f0 < n; f0 += sub)
    {
        const int64_t m = std::min<int64_t>(sub, n - f0);
        e->cur_call_last = f0 + m >= n;
        int rc = submit(e, 0, e->g.num_streams, m, d_xyz, d_intensity, d_poses, true, n, f0);
        if (rc)
            return rc;
    }
    return CC_OK;
}

int cc_engine_inputs_released(cc_engine* e, uint64_t* released_call, uint64_t* submitted_calls)
{
    if (!e)
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    // newest first: the events sit on one in-order stream, so the first one that has completed releases every older call as well
    uint64_t upto = e->released_seq;
    for (uint64_t q = e->call_seq; q > e->released_seq && q + cc_engine::REL_RING > e->call_seq; q--)
    {
        const int i = (int) (q % cc_engine::REL_RING);
        if (e->rel_recorded[i] != q)
            continue; // (its chains are still held back — deferred tail, lazy gate —, or it went through a path that only a synchronisation releases)
        if (hipEventQuery(e->ev_rel[i]) == hipSuccess)
        {
            upto = q;
            break;
        }
    }
    (void) hipGetLastError(); // (hipErrorNotReady of a query is not an error of the engine)
    int rc = note_released(e, upto);
    if (released_call)
        *released_call = e->released_seq;
    if (submitted_calls)
        *submitted_calls = e->call_seq;
    return rc;
}

int cc_engine_sync(cc_engine* e)
{
    if (!e)
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    int rc = finish_batch(e);
    if (rc)
        return rc;
    return first_stream_error(e, 0, e->g.num_streams);
}

void* cc_engine_hip_stream(cc_engine* e)
{
    if (e)
        (void) stop_resident(e); // (whoever asks for the stream is about to enqueue work on it)
    return e ? (void*) e->stream : nullptr;
}

int cc_engine_record_events(cc_engine* e, int enable)
{
    if (!e)
        return CC_ERR_INVALID_ARGUMENT;
    destroy_small_graphs(e);
    (void) hipSetDevice(e->device);
    int rc = stop_resident(e);
    if (rc)
        return rc;
    rc = finish_batch(e);
    if (rc)
        return rc;
    const int want = enable ? 1 : 0;
    if (want == e->g.record_events)
        return CC_OK;
    e->g.record_events = want;
    e->g.mirror_fields = want;
    const int cap = want ? 3 * (e->g.limit_columns + e->g.num_columns) + 4096 : 1;
    if (cap > e->g.event_capacity)
    {
        e->g.event_capacity = cap;
        rc = alloc_plane(e, &e->P.events, (size_t) e->g.num_streams * cap);
        if (rc)
            return rc;
    }
    return CC_OK;
}

int cc_engine_drain_events(cc_engine* e, int stream, cc_event* out, int64_t capacity, int64_t* n)
{
    if (!e || stream < 0 || stream >= e->g.num_streams || !n || capacity < 0 || (capacity > 0 && !out))
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    int rc = finish_batch(e);
    if (rc)
        return rc;
    auto& q = e->pending_events[stream];
    const int64_t k = std::min<int64_t>(capacity, (int64_t) q.size());
    if (k > 0)
        memcpy(out, q.data(), (size_t) k * sizeof(cc_event));
    q.erase(q.begin(), q.begin() + k);
    *n = k;
    return CC_OK;
}

int cc_engine_drain_links(cc_engine* e, int stream, int64_t* out, int64_t capacity, int64_t* n)
{
    if (!e || stream < 0 || stream >= e->g.num_streams || !n || capacity < 0 || (capacity > 0 && !out))
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    int rc = finish_batch(e);
    if (rc)
        return rc;
    auto& q = e->pending_links[stream];
    const int64_t k = std::min<int64_t>(capacity, (int64_t) q.size() / 4);
    if (k > 0)
        memcpy(out, q.data(), (size_t) k * 4 * sizeof(int64_t));
    q.erase(q.begin(), q.begin() + k * 4);
    *n = capacity == 0 ? (int64_t) q.size() / 4 : k;
    return CC_OK;
}

int cc_engine_pending_events(cc_engine* e, int stream, int64_t* n)
{
    if (!e || stream < 0 || stream >= e->g.num_streams || !n)
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    int rc = finish_batch(e);
    if (rc)
        return rc;
    *n = (int64_t) e->pending_events[stream].size();
    return CC_OK;
}

int cc_engine_stream_state(cc_engine* e, int stream, cc_stream_state* out)
{
    if (!e || stream < 0 || stream >= e->g.num_streams || !out)
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    int rc = finish_batch(e);
    if (rc)
        return rc;
    StreamState st;
    if (e->state_cached[stream])
        st = e->state_cache[stream];
    else
        CC_HIP_CHECK(e, hipMemcpy(&st, e->d_states + stream, sizeof(st), hipMemcpyDeviceToHost));
    fixup_overrun(e, stream, st);
    memset(out, 0, sizeof(*out));
    out->num_rows = e->g.num_rows;
    out->num_columns = e->g.num_columns;
    out->ring_buffer_max_columns = e->g.ring_cols;
    out->reset_required = st.reset_required;
    out->ring_buffer_start_global_column_index = st.ring_start;
    out->ring_buffer_end_global_column_index = st.ring_end;
    out->first_unfinished_global_column_index = st.first_unfinished;
    out->first_unpublished_global_column_index = st.first_unpublished;
    out->cluster_counter = st.cluster_counter;
    out->firings_consumed = st.firings_consumed;
    out->cells_published = st.cells_published;
    out->clusters_finished = st.clusters_finished;
    out->error = st.error;
    out->n_unfinished_trees = st.n_unfinished;
    out->error_a = st.error ? st.error_a : (int64_t) st.exceed_one_rotation;
    out->error_b = st.error ? st.error_b : (int64_t) st.serial_columns;
    return CC_OK;
}

// the columns of up to 8 ranges [from[i], to[i]], one after the other in the caller's arrays: from the views the last small call mirrored, else ONE launch
static int read_ranges(cc_engine* e, int stream, int nr, const int64_t* from, const int64_t* to, const cc_column_view* v)
{
    (void) hipSetDevice(e->device);
    int rc = finish_batch(e);
    if (rc)
        return rc;
    int64_t total = 0;
    for (int i = 0; i < nr; i++)
        total += to[i] - from[i] + 1;
    const size_t R = (size_t) e->g.num_rows;
    // Served from what the last small call mirrored with its results (HostMirror::view: the columns it segmented and the columns it published), if
    // that covers the request: no kernel, no copy engine, no synchronisation — the per-firing read of a front-end that keeps a mirror of range_image_
    if (e->small_view_ok && stream == e->small_view_stream && !v->number_of_child_points && total <= cck::MV_COLS)
    {
        const long long* hdr = e->h_small_view_hdr;
        // the views are written beside the call's results and stamped when they are complete: normally long before anybody asks
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; __atomic_load_n(&hdr[0], __ATOMIC_ACQUIRE) != (long long) e->small_seq_expected; spins++)
            if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
                break;
        const int nv = __atomic_load_n(&hdr[0], __ATOMIC_ACQUIRE) == (long long) e->small_seq_expected ? (int) hdr[1] : 0;
        int slot_of[cck::MV_COLS];
        bool all = nv > 0 && nv <= cck::MV_COLS;
        int k = 0;
        for (int i = 0; all && i < nr; i++)
            for (int64_t gcx = from[i]; all && gcx <= to[i]; gcx++)
            {
                int j = 0;
                while (j < nv && hdr[2 + j] != gcx)
                    j++;
                all = j < nv;
                slot_of[k++] = j;
            }
        if (all)
        {
            const cck::ViewOut m = cck::view_layout(e->h_small_view, (size_t) cck::MV_COLS * R);
            for (int c = 0; c < k; c++)
            {
                const size_t so = (size_t) slot_of[c] * R, dofs = (size_t) c * R;
#define VCOPY(dst, srcp, T)  \
    if (v->dst)              \
        memcpy(v->dst + dofs, (srcp) + so, R * sizeof(T));
                VCOPY(x, m.x, float) VCOPY(y, m.y, float) VCOPY(z, m.z, float) VCOPY(distance, m.dist, float) VCOPY(inclination_angle, m.incl, float);
                VCOPY(continuous_azimuth_angle, m.caz, double) VCOPY(global_column_index, m.gcol, int64_t) VCOPY(source_firing, m.src, int64_t);
                VCOPY(ground_point_label, m.ground, uint8_t) VCOPY(debug_ground_point_label, m.debug, uint8_t) VCOPY(is_ignored, m.ignored, uint8_t);
                VCOPY(id, m.id, uint64_t) VCOPY(tree_root_global_column, m.root_gcol, int64_t) VCOPY(tree_root_row, m.root_row, int32_t);
                VCOPY(finished_at_continuous_azimuth_angle, m.fin, double) VCOPY(tree_num_points, m.tpts, uint32_t) VCOPY(cluster_width, m.width, uint32_t);
                VCOPY(number_of_visited_neighbors, m.visits, int32_t) VCOPY(belongs_to_finished_cluster, m.finished, uint8_t);
                VCOPY(tree_parent_global_column, m.par_gcol, int64_t) VCOPY(tree_parent_row, m.par_row, int32_t);
#undef VCOPY
            }
            e->view_hits++;
            return CC_OK;
        }
    }
    e->view_misses++;
    const size_t n = (size_t) total * R;
    size_t used = 0;
    (void) cck::view_layout(nullptr, n, &used);
    const size_t bytes = used + 256;
    if (e->view_bytes < bytes)
    {
        void* p = nullptr;
        CC_HIP_CHECK(e, hipMalloc(&p, bytes));
        e->allocations.push_back(p);
        e->d_view = p;
        e->view_bytes = bytes;
    }
    if (e->h_view_bytes < bytes)
    {
        if (e->h_view)
            (void) hipHostFree(e->h_view);
        e->h_view = nullptr;
        e->h_view_bytes = 0;
        CC_HIP_CHECK(e, hipHostMalloc((void**) &e->h_view, bytes));
        e->h_view_bytes = bytes;
    }
    char* base = (char*) e->d_view;
    cck::ViewOut o = cck::view_layout(base, n);
    char* const nchild_at = (char*) o.nchild;
    // the optional fields cost a child-count pass and extra copies: only when the caller asked for one of them
    if (!v->number_of_child_points)
        o.nchild = nullptr;
    int max_back = e->cfg.max_steps_in_row < e->g.ring_cols - 1 ? e->cfg.max_steps_in_row : e->g.ring_cols - 1;
    max_back = max_back < 0 ? 0 : (max_back > 255 ? 255 : max_back);
    cck::ViewRanges vr;
    vr.n = nr;
    int acc = 0;
    for (int i = 0; i < 8; i++)
    {
        vr.start[i] = acc;
        vr.from[i] = i < nr ? (long long) from[i] : 0;
        if (i < nr)
            acc += (int) (to[i] - from[i] + 1);
    }
    vr.start[8] = acc;
    hipLaunchKernelGGL(cck::k_view, dim3((unsigned) total), dim3(64), 0, query_stream(e), e->g, e->P, e->d_states, stream, vr, o, max_back);
    CC_HIP_CHECK(e, hipGetLastError());
    // one copy of the whole staging block (a call per field costs more than the bytes for the few columns a live mirror reads)
    CC_HIP_CHECK(e, hipMemcpyAsync(e->h_view, base, used, hipMemcpyDeviceToHost, query_stream(e)));
    CC_HIP_CHECK(e, hipStreamSynchronize(query_stream(e)));
#define COPY(dst, srcp, T)                                                                        \
    if (v->dst)                                                                                   \
        memcpy(v->dst, e->h_view + ((const char*) (srcp) - base), n * sizeof(T));
    COPY(x, o.x, float) COPY(y, o.y, float) COPY(z, o.z, float) COPY(distance, o.dist, float) COPY(inclination_angle, o.incl, float);
    COPY(continuous_azimuth_angle, o.caz, double) COPY(global_column_index, o.gcol, int64_t) COPY(source_firing, o.src, int64_t);
    COPY(ground_point_label, o.ground, uint8_t) COPY(debug_ground_point_label, o.debug, uint8_t) COPY(is_ignored, o.ignored, uint8_t);
    COPY(id, o.id, uint64_t) COPY(tree_root_global_column, o.root_gcol, int64_t) COPY(tree_root_row, o.root_row, int32_t);
    COPY(finished_at_continuous_azimuth_angle, o.fin, double) COPY(tree_num_points, o.tpts, uint32_t) COPY(cluster_width, o.width, uint32_t);
    COPY(number_of_visited_neighbors, o.visits, int32_t) COPY(belongs_to_finished_cluster, o.finished, uint8_t);
    COPY(tree_parent_global_column, o.par_gcol, int64_t) COPY(tree_parent_row, o.par_row, int32_t);
    if (v->number_of_child_points)
        memcpy(v->number_of_child_points, e->h_view + (nchild_at - base), n * sizeof(uint32_t));
#undef COPY
    return CC_OK;
}

int cc_engine_read_columns(cc_engine* e, int stream, int64_t from, int64_t to, const cc_column_view* v)
{
    if (!e || stream < 0 || stream >= e->g.num_streams || !v || to < from || to - from >= e->g.ring_cols)
        return CC_ERR_INVALID_ARGUMENT;
    return read_ranges(e, stream, 1, &from, &to, v);
}

int cc_engine_read_column_ranges(cc_engine* e, int stream, int n_ranges, const int64_t* from, const int64_t* to, const cc_column_view* v)
{
    if (!e || stream < 0 || stream >= e->g.num_streams || !v || n_ranges < 0 || n_ranges > 8 || (n_ranges > 0 && (!from || !to)))
        return CC_ERR_INVALID_ARGUMENT;
    int64_t total = 0;
    for (int i = 0; i < n_ranges; i++)
    {
        if (to[i] < from[i])
            return CC_ERR_INVALID_ARGUMENT;
        total += to[i] - from[i] + 1;
    }
    if (total > e->g.ring_cols)
        return CC_ERR_INVALID_ARGUMENT;
    if (total == 0)
        return CC_OK;
    return read_ranges(e, stream, n_ranges, from, to, v);
}

int cc_engine_gather_cluster_points(cc_engine* e, int stream, int64_t n, const uint32_t* cluster_ids, const int64_t* col_from,
                                    const int64_t* col_to, const uint32_t* n_points, int64_t* h_gcol, int32_t* h_row)
{
    if (!e || stream < 0 || stream >= e->g.num_streams || n < 0 || (n > 0 && (!cluster_ids || !col_from || !col_to || !n_points)))
        return CC_ERR_INVALID_ARGUMENT;
    if (n == 0)
        return CC_OK;
    (void) hipSetDevice(e->device);
    int rc = finish_batch(e);
    if (rc)
        return rc;
    std::vector<long long> offs((size_t) n);
    long long total = 0;
    for (int64_t i = 0; i < n; i++)
    {
        offs[(size_t) i] = total;
        total += n_points[i];
    }
    if (total > 0 && (!h_gcol || !h_row))
        return CC_ERR_INVALID_ARGUMENT;
    // one device block: descriptors (cid, n_points: u32; from, to, offset: i64), mismatch counter, outputs
    const size_t desc = (size_t) n * (4 + 4 + 8 + 8 + 8) + 64, outb = (size_t) total * (8 + 4) + 64;
    // grow-only scratch owned by the engine (a hipMalloc / hipFree pair per call costs more than the gather)
    if (e->gather_bytes < desc + outb)
    {
        void* p = nullptr;
        CC_HIP_CHECK(e, hipMalloc(&p, (desc + outb) * 2));
        e->allocations.push_back(p);
        e->d_gather = (char*) p;
        e->gather_bytes = (desc + outb) * 2;
    }
    char* d = e->d_gather;
    auto fail = [&](int code) { return code; };
    cck::ClusterQuery q;
    char* b = d;
    q.col_from = (const long long*) b;
    b += (size_t) n * 8;
    q.col_to = (const long long*) b;
    b += (size_t) n * 8;
    q.offset = (const long long*) b;
    b += (size_t) n * 8;
    q.cid = (const unsigned*) b;
    b += (size_t) n * 4;
    q.n_points = (const unsigned*) b;
    b += (size_t) n * 4;
    b = d + ((b - d + 63) / 64) * 64;
    q.mismatch = (int*) b;
    b += 64;
    q.out_gcol = (long long*) b;
    b += (size_t) total * 8;
    q.out_row = (int*) b;
    if (hipMemcpyAsync((void*) q.col_from, col_from, (size_t) n * 8, hipMemcpyHostToDevice, query_stream(e)) != hipSuccess ||
        hipMemcpyAsync((void*) q.col_to, col_to, (size_t) n * 8, hipMemcpyHostToDevice, query_stream(e)) != hipSuccess ||
        hipMemcpyAsync((void*) q.offset, offs.data(), (size_t) n * 8, hipMemcpyHostToDevice, query_stream(e)) != hipSuccess ||
        hipMemcpyAsync((void*) q.cid, cluster_ids, (size_t) n * 4, hipMemcpyHostToDevice, query_stream(e)) != hipSuccess ||
        hipMemcpyAsync((void*) q.n_points, n_points, (size_t) n * 4, hipMemcpyHostToDevice, query_stream(e)) != hipSuccess ||
        hipMemsetAsync(q.mismatch, 0, 64, query_stream(e)) != hipSuccess)
    {
        e->error = "cc_engine_gather_cluster_points: copy failed";
        return fail(CC_ERR_HIP);
    }
    hipLaunchKernelGGL(cck::k_gather_clusters, dim3((unsigned) n), dim3(64), 0, query_stream(e), e->g, e->P, e->d_states, stream, q);
    int mismatch = 0;
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&mismatch, q.mismatch, 4, hipMemcpyDeviceToHost, query_stream(e)) != hipSuccess ||
        (total > 0 && (hipMemcpyAsync(h_gcol, q.out_gcol, (size_t) total * 8, hipMemcpyDeviceToHost, query_stream(e)) != hipSuccess ||
                       hipMemcpyAsync(h_row, q.out_row, (size_t) total * 4, hipMemcpyDeviceToHost, query_stream(e)) != hipSuccess)) ||
        hipStreamSynchronize(query_stream(e)) != hipSuccess)
    {
        e->error = "cc_engine_gather_cluster_points: launch / copy failed";
        return fail(CC_ERR_HIP);
    }
    if (mismatch != 0)
    {
        e->error = "cc_engine_gather_cluster_points: " + std::to_string(mismatch) + " cluster descriptor(s) do not match the engine state";
        return CC_ERR_INVALID_ARGUMENT;
    }
    return CC_OK;
}

int cc_engine_output_planes(cc_engine* e, int stream, const uint8_t** d_ground_label, const uint32_t** d_cluster_id)
{
    if (!e || stream < 0 || stream >= e->g.num_streams)
        return CC_ERR_INVALID_ARGUMENT;
    if (d_ground_label)
        *d_ground_label = e->P.ground + (size_t) stream * e->g.cells;
    if (d_cluster_id)
        *d_cluster_id = e->P.id + (size_t) stream * e->g.cells;
    return CC_OK;
}

int cc_engine_set_option(cc_engine* e, const char* name, int64_t value)
{
    if (!e || !name)
        return CC_ERR_INVALID_ARGUMENT;
    destroy_small_graphs(e);
    (void) hipSetDevice(e->device);
    int rc = stop_resident(e); // (the resident kernel bakes options, geometry and plane pointers in, like a captured graph)
    if (rc)
        return rc;
    rc = finish_batch(e);
    if (rc)
        return rc;
    const std::string n(name);
    if (n == "resident")
        e->resident_opt = value != 0;
    else if (n == "mirror_views")
        e->mirror_views = value != 0;
    else if (n == "resident_idle_ms")
        e->res_idle_ms = (int) std::max<int64_t>(1, std::min<int64_t>(value, 10000));
    else if (n == "debug_flags")
        e->g.debug_flags = (int32_t) value;
    else if (n == "lds_tree_limit")
        e->g.lds_tree_limit = (int32_t) (value < 1 ? 1 : (value > TREE_SLOTS ? TREE_SLOTS : value));
    else if (n == "pipeline")
    {
        e->allow_pipeline = value != 0; // 0: one stream, 1: three chains, 2: four (window scan on its own stream)
        e->pipeline_depth = value >= 2 ? 2 : 1;
    }
    else if (n == "graphs")
        e->allow_graphs = value != 0;
    else if (n == "sub_batch")
        e->sub_batch = value < 0 ? 0 : value;
    else if (n == "table_on_insert_chain")
        e->table_on_insert_chain = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "assoc_sweep_blocks")
        e->assoc_sweep_blocks = value < 1 ? 1 : (value > 1024 ? 1024 : (int) value);
    else if (n == "forget_inclination_table")
    {
        // sc_inclination_angles_between_lasers_ as a freshly constructed object has it (cc_engine_reset keeps it, like the reference's resize does):
        // for callers that ran made-up data through a new engine (the drop-in class's warm-up)
        if (value != 0)
            CC_HIP_CHECK(e, hipMemset(e->P.curtab, 0xFF, (size_t) e->g.num_streams * (size_t) e->g.num_rows * sizeof(float)));
    }
    else if (n == "prewarm_small_graphs")
    {
        // Capturing and instantiating the graph of a call size takes milliseconds, once per size: a front-end that feeds a live sensor pays them
        // here (the drop-in class does, in reset()), not in front of the first call of every size — with the asynchronous mode handing the engine
        // whatever has queued up (1 .. 8 firings), eight such stalls of up to 30 ms each sat inside the first seconds of a stream
        if (value != 0 && e->g.num_streams == 1)
        {
            // buffers first, sized for the largest call the drop-in class makes (a rotation of firings, when its caller ran ahead): growing them
            // later re-allocates what the captured graphs point at, i.e. drops the graphs again
            const size_t cap = (size_t) std::max(4096, e->g.num_columns);
            if (e->ego_capacity < cap)
            {
                bool ok = true;
                for (int i = 0; i < 4 && ok; i++)
                    ok = alloc_plane(e, &e->d_ego[i], cap * cck::EGO_STRIDE) == CC_OK;
                if (ok)
                {
                    e->ego_capacity = cap;
                    e->small_graphs_stale = true;
                }
            }
            (void) ensure_prep(e, (size_t) e->g.num_columns * (size_t) e->g.num_rows);
            // the grow-only scratch of cc_engine_read_columns (a mirror reads a few columns per call; a device and a pinned host block) and of
            // cc_engine_gather_cluster_points: a pinned allocation in front of the first finished cluster is a stall of milliseconds
            {
                const size_t vb = (size_t) 512 * e->g.num_rows * (5 * 4 + 8 + 3 * 8 + 3 + 8 + 4 + 8 + 8 + 5 * 4 + 1) + 256;
                void* pv = nullptr;
                if (e->view_bytes < vb && hipMalloc(&pv, vb) == hipSuccess)
                {
                    e->allocations.push_back(pv);
                    e->d_view = pv;
                    e->view_bytes = vb;
                }
                if (e->h_view_bytes < vb)
                {
                    if (e->h_view)
                        (void) hipHostFree(e->h_view);
                    e->h_view = nullptr;
                    e->h_view_bytes = 0;
                    if (hipHostMalloc((void**) &e->h_view, vb) == hipSuccess)
                        e->h_view_bytes = vb;
                }
                const size_t gb = (size_t) 4 << 20;
                void* pg = nullptr;
                if (e->gather_bytes < gb && hipMalloc(&pg, gb) == hipSuccess)
                {
                    e->allocations.push_back(pg);
                    e->d_gather = (char*) pg;
                    e->gather_bytes = gb;
                }
            }
            for (int64_t k = 1; k <= SMALL_MAX; k++)
                (void) add_firings_small(e, 0, k, nullptr, nullptr, nullptr);
        }
    }
    else if (n == "defer_tail_max_streams")
        e->defer_tail_max_streams = value < 0 ? 0 : (int) value;
    else if (n == "assoc_cooldown")
        e->bail_cooldown_batches = value < 0 ? 0 : (value > 1000 ? 1000 : (int) value);
    else if (n == "ego_on_insert_chain")
        e->ego_on_insert_chain = value != 0;
    else if (n == "debug_no_assoc_fallback")
        e->debug_no_assoc_fallback = value != 0;
    else if (n == "publish_off_chain")
        e->publish_off_chain = value != 0;
    else if (n == "timing_every")
        e->timing_every = value < 1 ? 1 : (int) value;
    else if (n == "parallel_insert")
    {
        e->parallel_insert = value != 0;
        e->parallel_insert_multi = value == 1;
    }
    else if (n == "input_on_engine_stream")
        e->input_on_engine_stream = value != 0;
    else if (n == "assoc_waves")
    {
        // 1: k_assoc_lds, 3: k_assoc3 without the links wave, 4: with it, 0 (default): k_assoc3, links wave up to 256 streams (2: as 3)
        e->assoc_waves_auto = value <= 0 || value > 4;
        e->assoc_waves = e->assoc_waves_auto ? 3 : (int) value;
    }
    else if (n == "insert_split_blocks")
        e->insert_split_blocks = (int) (value < 0 ? 0 : (value > 8 ? 8 : value));
    else if (n == "insert_narrow_blocks")
        e->insert_narrow_blocks = value < 0 ? 0 : (value > 8 ? 8 : (int) value);
    else if (n == "insert_wide_max_streams")
        e->insert_wide_max_streams = value < 0 ? 0 : (value > (1 << 20) ? (1 << 20) : (int) value);
    else if (n == "skip_idle_fallbacks")
        e->skip_idle_fallbacks = value != 0;
    else if (n == "fuse_front")
        e->fuse_front = value != 0;
    else if (n == "small_front")
    {
        e->small_front = value != 0;
        e->small_graphs_stale = true;
    }
    else if (n == "small_all")
    {
        e->small_all = value != 0;
        e->small_graphs_stale = true;
    }
    else if (n == "small_direct")
        e->small_direct = value != 0;
    else if (n == "insert_lds_pad")
        e->insert_lds_pad_kb = (int) std::max<int64_t>(0, std::min<int64_t>(value, 120));
    else if (n == "check_input_lifetime")
    {
        int rcf = finish_batch(e);
        if (rcf)
            return rcf;
        e->check_input_lifetime = (int) std::max<int64_t>(0, std::min<int64_t>(value, 2));
        e->live_inputs.clear();
    }
    else if (n == "lazy_gate")
    {
        int rcf = flush_deferred(e);
        if (rcf)
            return rcf;
        e->lazy_gate_max_streams = (int) std::max<int64_t>(0, std::min<int64_t>(value, 4096));
        e->lazy_ok = true;
        e->lazy_miss = 0;
    }
    else if (n == "lazy_gate_from")
    {
        int rcf = flush_deferred(e);
        if (rcf)
            return rcf;
        e->lazy_gate_from_streams = (int) std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20));
        e->lazy_ok = true;
        e->lazy_miss = 0;
    }
    else if (n == "seg_small_max")
    {
        e->seg_small_max = value < 0 ? 0 : (value > 63 ? 63 : (int) value);
        e->small_graphs_stale = true;
    }
    else if (n == "assoc_batch")
        e->assoc_batch = value != 0;
    else if (n == "assoc_rounds")
        e->assoc_rounds = (int) (value < 0 ? 0 : (value > 8 ? 8 : value));
    else if (n == "ego_off_chain")
        e->ego_off_chain = value != 0;
    else if (n == "scan_store_fin")
        e->scan_store_fin = value < 0 ? -1 : (value ? 1 : 0);
    else if (n == "scan_split")
        e->scan_split = (int) std::max<int64_t>(0, std::min<int64_t>(value, 2));
    else if (n == "scan_cap")
        e->g.scan_cap = (int) std::max<int64_t>(1, std::min<int64_t>(value, 1 << 20));
    else if (n == "scan_long_records")
        e->g.sl_cap = (int) std::max<int64_t>(1, std::min<int64_t>(value, cck::SL_CAP));
    else if (n == "scan_packed")
    {
        e->scan_packed = value < 0 ? -1 : (value != 0 ? 1 : 0);
    }
    else if (n == "mirror_fields")
        e->g.mirror_fields = value != 0;
    else if (n == "limit_columns")
        e->g.limit_columns = (int32_t) (value < 1 ? 1 : value);
    else
    {
        e->error = "unknown option " + n;
        return CC_ERR_INVALID_ARGUMENT;
    }
    return CC_OK;
}

int cc_engine_enable_timing(cc_engine* e, int enable)
{
    if (!e)
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    int rc = stop_resident(e);
    if (rc)
        return rc;
    rc = finish_batch(e);
    if (rc)
        return rc;
    e->timing = enable != 0;
    for (double& v : e->kernel_ms)
        v = 0;
    e->kernel_launches = 0;
    e->timing_pass = 0;
    e->prep_ahead_ms = 0;
    return CC_OK;
}

int cc_engine_kernel_times(cc_engine* e, double ms[7], uint64_t* launches)
{
    if (!e || !ms)
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    int rc = finish_batch(e);
    if (rc)
        return rc;
    // timing_every > 1: the sums of the sampled passes, scaled to all passes (average duration per launch x launches)
    const bool sampled = e->timing_every > 1 && e->kernel_launches > 0 && e->timing_pass > e->kernel_launches;
    const double scale = sampled ? (double) e->timing_pass / (double) e->kernel_launches : 1.0;
    for (int k = 0; k < 7; k++)
        ms[k] = e->kernel_ms[k] * scale + (k == 0 ? e->prep_ahead_ms : 0.0);
    if (launches)
        *launches = sampled ? e->timing_pass : e->kernel_launches;
    return CC_OK;
}

int cc_engine_totals(cc_engine* e, uint64_t* cells_published, uint64_t* clusters_finished, uint64_t* firings_consumed,
                     uint64_t* serial_columns)
{
    if (!e)
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    int rc = finish_batch(e);
    if (rc)
        return rc;
    std::vector<StreamState> st(e->g.num_streams);
    CC_HIP_CHECK(e, hipMemcpy(st.data(), e->d_states, st.size() * sizeof(StreamState), hipMemcpyDeviceToHost));
    uint64_t a = 0, b = 0, c = 0, d = 0;
    for (auto& s : st)
    {
        a += s.cells_published;
        b += s.clusters_finished;
        c += s.firings_consumed;
        d += s.serial_columns;
    }
    if (cells_published)
        *cells_published = a;
    if (clusters_finished)
        *clusters_finished = b;
    if (firings_consumed)
        *firings_consumed = c;
    if (serial_columns)
        *serial_columns = d;
    return CC_OK;
}

// ---- frame scatter of a replayed sequence (kitti_demo.cpp:173-224), include/cc_kitti.h ------------------------------------------------
int cc_engine_scatter_info(cc_engine* e, int n, const int32_t* streams, const int64_t* from, const int64_t* to, const int32_t* d_original_index,
                           int slots, int32_t* h_min_frame, int32_t* h_max_frame)
{
    if (!e || n < 0 || slots < 1 || (n > 0 && (!streams || !from || !to || !d_original_index || !h_min_frame || !h_max_frame)))
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    int rc = stop_resident(e);
    if (rc)
        return rc;
    rc = finish_batch(e);
    if (rc)
        return rc;
    size_t total = 0;
    for (int i = 0; i < n; i++)
    {
        if (streams[i] < 0 || streams[i] >= e->g.num_streams || from[i] < 0 || to[i] - from[i] >= e->g.ring_cols)
            return CC_ERR_INVALID_ARGUMENT; // (columns outside the stream's published ring window read as empty: the kernels check)
        if (to[i] >= from[i])
            total += (size_t) (to[i] - from[i] + 1);
    }
    if (total == 0)
        return CC_OK;
    const size_t need = total * 8 + 64;
    if (e->gather_bytes < need)
    {
        void* p = nullptr;
        CC_HIP_CHECK(e, hipMalloc(&p, need * 2));
        e->allocations.push_back(p);
        e->d_gather = (char*) p;
        e->gather_bytes = need * 2;
    }
    int* d_min = (int*) e->d_gather;
    int* d_max = d_min + total;
    size_t o = 0;
    for (int i = 0; i < n; i++)
        if (to[i] >= from[i])
        {
            const unsigned cols = (unsigned) (to[i] - from[i] + 1);
            hipLaunchKernelGGL(cck::k_scatter_info, dim3(cols), dim3(64), 0, e->stream, e->g, e->P, (const StreamState*) e->d_states, streams[i], (long long) from[i], d_original_index,
                               slots, d_min + o, d_max + o);
            o += cols;
        }
    CC_HIP_CHECK(e, hipGetLastError());
    CC_HIP_CHECK(e, hipMemcpyAsync(h_min_frame, d_min, total * 4, hipMemcpyDeviceToHost, e->stream));
    CC_HIP_CHECK(e, hipMemcpyAsync(h_max_frame, d_max, total * 4, hipMemcpyDeviceToHost, e->stream));
    CC_HIP_CHECK(e, hipStreamSynchronize(e->stream));
    return CC_OK;
}

int cc_engine_scatter_apply(cc_engine* e, int stream, int64_t from, int64_t to, const int32_t* d_original_index, int slots, uint8_t* d_is_ground,
                            uint32_t* d_detection, int64_t max_points)
{
    if (!e || stream < 0 || stream >= e->g.num_streams || slots < 1 || !d_original_index || !d_is_ground || !d_detection || max_points < 1 ||
        from < 0 || to - from >= e->g.ring_cols)
        return CC_ERR_INVALID_ARGUMENT;
    if (to < from)
        return CC_OK;
    (void) hipSetDevice(e->device);
    int rc = stop_resident(e);
    if (rc)
        return rc;
    rc = finish_batch(e);
    if (rc)
        return rc;
    hipLaunchKernelGGL(cck::k_scatter_apply, dim3((unsigned) (to - from + 1)), dim3(64), 0, e->stream, e->g, e->P, (const StreamState*) e->d_states, stream, (long long) from,
                       d_original_index, slots, d_is_ground, d_detection, (long long) max_points);
    CC_HIP_CHECK(e, hipGetLastError());
    return CC_OK; // (asynchronous on cc_engine_hip_stream(e): cc_eval_frame_device on the same stream, or cc_engine_sync, orders behind it)
}

int cc_engine_view_counters(cc_engine* e, uint64_t* served_from_mirror, uint64_t* served_by_kernel)
{
    if (!e)
        return CC_ERR_INVALID_ARGUMENT;
    if (served_from_mirror)
        *served_from_mirror = e->view_hits;
    if (served_by_kernel)
        *served_by_kernel = e->view_misses;
    return CC_OK;
}

int cc_engine_resident_counters(cc_engine* e, uint64_t* launches, uint64_t* calls, int* running)
{
    if (!e)
        return CC_ERR_INVALID_ARGUMENT;
    if (launches)
        *launches = e->res_launches;
    if (calls)
        *calls = e->res_calls + (e->res_running ? __atomic_load_n(&e->h_res_ctl->calls, __ATOMIC_ACQUIRE) : 0ull);
    if (running)
        *running = (e->res_running && __atomic_load_n(&e->h_res_ctl->exited, __ATOMIC_ACQUIRE) == 0ull) ? 1 : 0;
    return CC_OK;
}

int cc_engine_gate_counters(cc_engine* e, uint64_t* lazy_batches, uint64_t* lazy_redone)
{
    if (!e)
        return CC_ERR_INVALID_ARGUMENT;
    if (lazy_batches)
        *lazy_batches = e->lazy_batches;
    if (lazy_redone)
        *lazy_redone = e->lazy_redone;
    return CC_OK;
}

int cc_engine_batch_counters(cc_engine* e, uint64_t* batch_columns, uint64_t* batch_bails, uint64_t bail_reasons[8])
{
    if (!e)
        return CC_ERR_INVALID_ARGUMENT;
    (void) hipSetDevice(e->device);
    int rc = finish_batch(e);
    if (rc)
        return rc;
    std::vector<StreamState> st(e->g.num_streams);
    CC_HIP_CHECK(e, hipMemcpy(st.data(), e->d_states, st.size() * sizeof(StreamState), hipMemcpyDeviceToHost));
    uint64_t a = 0, b = 0;
    if (bail_reasons)
        for (int i = 0; i < 8; i++)
            bail_reasons[i] = 0;
    for (auto& s : st)
    {
        a += s.batch_columns;
        b += s.batch_bails;
        if (bail_reasons)
            for (int i = 0; i < 8; i++)
                bail_reasons[i] += s.batch_bail_reason[i];
    }
    if (bail_reasons)
        bail_reasons[7] = e->small_tail_launches; // (not a reason: small calls whose serial fall-backs the host launched behind k_small_all)
    if (batch_columns)
        *batch_columns = a;
    if (batch_bails)
        *batch_bails = b;
    return CC_OK;
}

int cc_engine_debug_counters(cc_engine* e, int stream, uint64_t out[16])
{
    if (!e || stream < 0 || stream >= e->g.num_streams || !out)
        return CC_ERR_INVALID_ARGUMENT;
    StreamState st;
    CC_HIP_CHECK(e, hipMemcpy(&st, e->d_states + stream, sizeof(st), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; i++)
        out[i] = st.dbg[i];
    return CC_OK;
}

const char* cc_engine_last_error(cc_engine* e)
{
    return e ? e->error.c_str() : "null engine";
}

} // extern "C"
