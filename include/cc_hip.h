/* cc_hip.h — C-ABI of the MI355X (gfx950) continuous-clustering hot path.
 *
 * This is the drop-in boundary UNDER the reference's C++ class
 * continuous_clustering::ContinuousClustering (include/continuous_clustering/clustering/
 * continuous_clustering.hpp:197-290 of UniBwTAS/continuous_clustering): plain pointers, sizes and
 * int status codes; no C++ types, no exceptions, no torch types. The C++ class in
 * continuous_clustering_amd/csrc/continuous_clustering.hpp re-creates the reference API on top of
 * it (INTEGRATION.md shows the binding a reference maintainer would add).
 *
 * Vocabulary follows the reference: a *firing* is one vertical set of num_rows laser returns, a
 * *column* is one range-image column (global column index = int64 that grows forever, local column
 * = global % ring_buffer_max_columns), a *sensor stream* is what one ContinuousClustering object
 * consumes. One cc_engine owns `num_streams` independent sensor streams of identical geometry and
 * configuration on one GPU and advances all of them with batched kernel launches.
 *
 * Reference entry point replaced by each function is cited as cc.cpp:<line> =
 * src/clustering/continuous_clustering.cpp, cc.hpp:<line> = the class header above.
 */
#ifndef CC_HIP_H
#define CC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (the C++ wrapper turns the CC_ERR_* of a stream into the reference's
 *      std::runtime_error texts, cc.cpp:90-91, :298-299, :337-344, :1052, :1073-1075) ---------- */
enum
{
    CC_OK = 0,
    CC_ERR_INVALID_ARGUMENT = 1,
    CC_ERR_HIP = 2,                 /* a HIP runtime call failed; see cc_engine_last_error */
    CC_ERR_NO_DEVICE = 3,           /* no gfx950 device visible: the product path never falls back to CPU */
    CC_ERR_FIRING_SIZE = 4,         /* cc.cpp:90-91 */
    CC_ERR_NO_ROBOT_TRANSFORM = 5,  /* cc.cpp:298-299 */
    CC_ERR_RING_OVERRUN = 6,        /* cc.cpp:337-344 "This column is not cleared ..." */
    CC_ERR_BOOKKEEPING = 7,         /* cc.cpp:1052 / :1073-1075 */
    CC_ERR_CAPACITY = 8,            /* an engine-side pool (unfinished trees, events) is full */
    CC_ERR_NEGATIVE_COLUMN = 9      /* reference computes a negative global column (undefined behaviour there) */
};

/* ---- ground / debug label values (cc.hpp:15-22, general.hpp:208-357) ----------------------- */
enum
{
    CC_GP_UNKNOWN = 143,     /* WHITE */
    CC_GP_GROUND = 54,       /* GREEN */
    CC_GP_OBSTACLE = 119,    /* RED */
    CC_GP_EGO_VEHICLE = 85,  /* MAGENTA */
    CC_GP_FOG = 71,          /* LIGHTGRAY */
    CC_DBG_GRAY = 53,
    CC_DBG_ORANGE = 105,
    CC_DBG_GREEN = 54,
    CC_DBG_YELLOWGREEN = 146,
    CC_DBG_YELLOW = 145,
    CC_DBG_RED = 119,
    CC_DBG_DARKRED = 32,
    CC_DBG_VIOLET = 141,
    CC_DBG_LIGHTGRAY = 71,
    CC_DBG_WHITE = 143
};

/* ---- configuration: every field of continuous_clustering::Configuration (cc.hpp:24-87) as
 *      fixed-width scalars, same names, same defaults (cc_config_default) ----------------------- */
typedef struct cc_config
{
    /* GeneralConfiguration cc.hpp:24-27 */
    int32_t is_single_threaded;
    /* ContinuousRangeImageConfiguration cc.hpp:29-34 */
    int32_t sensor_is_clockwise;
    int32_t num_columns;
    int32_t supplement_inclination_angle_for_nan_cells;
    /* ContinuousGroundSegmentationConfiguration cc.hpp:36-66 */
    float max_slope;
    float first_ring_as_ground_max_allowed_z_diff;
    float first_ring_as_ground_min_allowed_z_diff;
    float last_ground_point_slope_higher_than;
    float last_ground_point_distance_smaller_than;
    float ground_because_close_to_last_certain_ground_max_z_diff;
    float ground_because_close_to_last_certain_ground_max_dist_diff;
    float obstacle_because_next_certain_obstacle_max_dist_diff;
    int32_t use_terrain;
    float terrain_max_allowed_z_diff;
    float height_ref_to_maximum_;
    float height_ref_to_ground_;
    float length_ref_to_front_end_;
    float length_ref_to_rear_end_;
    float width_ref_to_left_mirror_;
    float width_ref_to_right_mirror_;
    int32_t fog_filtering_enabled;
    int32_t fog_filtering_intensity_below; /* uint8_t in the reference */
    float fog_filtering_distance_below;
    float fog_filtering_inclination_above;
    /* ContinuousClusteringConfiguration cc.hpp:68-79 */
    float max_distance;
    int32_t max_steps_in_row;
    int32_t max_steps_in_column;
    int32_t stop_after_association_enabled;
    int32_t stop_after_association_min_steps;
    int32_t ignore_points_in_chessboard_pattern;
    int32_t ignore_points_with_too_big_inclination_angle_diff;
    int32_t use_last_point_for_cluster_stamp;
    int32_t cluster_point_trees_every_nth_column;
} cc_config;

/* Library defaults of cc.hpp:24-79. */
void cc_config_default(cc_config* cfg);
/* The KITTI parameter set of src/tools/kitti_demo.cpp:279-294 (single threaded, 2200 columns,
 * max_distance 0.5, chessboard off, ego box +-3 / +-1.5 / +0.5 / -1.7). */
void cc_config_kitti(cc_config* cfg);

/* ---- events: what the reference reports through its two std::function callbacks, in the order
 *      the single-threaded reference would have invoked them ------------------- */
enum
{
    CC_EV_GROUND_COLUMN = 1, /* finished_column_callback_(a, a, true)           cc.cpp:618-620  */
    CC_EV_CLUSTER = 2,       /* a finished cluster that received an id           cc.cpp:936-940,
                                a,b = first/last global column it covers, c = id, d = number of points;
                                the reference invokes finished_cluster_callback_ iff d > 20 (cc.cpp:1023) */
    CC_EV_PUBLISH_COLUMNS = 3 /* finished_column_callback_(a, b, false), may be empty (b < a) cc.cpp:1087-1089 */
};

typedef struct cc_event
{
    int32_t type;
    int32_t stream;
    int64_t a;
    int64_t b;
    uint32_t c;
    uint32_t d;
    int64_t column; /* global column whose processing produced the event */
} cc_event;

/* ---- per-stream scalar state readable by the caller (public members cc.hpp:244-251 plus the
 *      counters the front-ends derive) --------------------------------------------------------- */
typedef struct cc_stream_state
{
    int32_t num_rows;                          /* num_rows_                       cc.hpp:248 */
    int32_t num_columns;                       /* num_columns_                    cc.hpp:247 */
    int32_t ring_buffer_max_columns;           /* ring_buffer_max_columns         cc.hpp:246 */
    int32_t reset_required;                    /* resetRequired()                 cc.cpp:83-86 */
    int64_t ring_buffer_start_global_column_index; /* cc.hpp:250 */
    int64_t ring_buffer_end_global_column_index;   /* cc.hpp:251 */
    int64_t first_unfinished_global_column_index;  /* srig_first_unfinished_..., columns below are segmented */
    int64_t first_unpublished_global_column_index; /* sc_first_unpublished_..., columns below are published */
    uint64_t cluster_counter;                  /* sc_cluster_counter_ (next id)   cc.hpp:274 */
    uint64_t firings_consumed;                 /* firings inserted since reset */
    uint64_t cells_published;                  /* num_rows * published columns since reset */
    uint64_t clusters_finished;                /* clusters that received an id since reset */
    int32_t error;                             /* CC_OK or CC_ERR_* raised inside a kernel for this stream */
    int32_t n_unfinished_trees;
    int64_t error_a;                           /* operands of the reference's error text */
    int64_t error_b;
} cc_stream_state;

/* ---- host-side view of range-image columns (the fields of continuous_clustering::Point,
 *      cc.hpp:126-161, that the algorithm owns). Every pointer may be NULL (field not wanted).
 *      Arrays are column-major like the reference's range_image_ (cc.cpp:181): element
 *      [(col - from) * num_rows + row]. ------------------------------------------------------- */
typedef struct cc_column_view
{
    float* x;                        /* Point::xyz (odom frame) */
    float* y;
    float* z;
    float* distance;                 /* Point::distance */
    float* inclination_angle;        /* Point::inclination_angle (NaN cells supplemented, cc.cpp:364-369) */
    double* continuous_azimuth_angle;/* Point::continuous_azimuth_angle */
    int64_t* global_column_index;    /* Point::global_column_index (-1 for never-segmented cleared cells) */
    int64_t* source_firing;          /* sequence number (since reset) of the firing whose point fills the cell, -1 if empty;
                                        lets the host attach stamp / firing_index / globally_unique_point_index /
                                        intensity / azimuth_angle, which never leave the host */
    uint8_t* ground_point_label;     /* Point::ground_point_label */
    uint8_t* debug_ground_point_label;
    uint8_t* is_ignored;             /* Point::is_ignored */
    uint64_t* id;                    /* Point::id — cluster id, reference numbering (cc.cpp:939), 0 = none */
    int64_t* tree_root_global_column;/* global column of Point::tree_root_, -1 = none */
    int32_t* tree_root_row;          /* row of Point::tree_root_ */
    /* The remaining clustering fields of Point that the reference's ROS packers read (ros_utils.cpp:289-295) and that
     * collectPointsForCusterAndPublish walks (cc.cpp:996-1016). Final once the point's tree is finished, i.e. for every
     * published column. Produced when the engine option "mirror_fields" is on (default for 1 stream). */
    double* finished_at_continuous_azimuth_angle; /* Point::finished_at_continuous_azimuth_angle: tree roots, 0 elsewhere (cc.cpp:669,818) */
    uint32_t* tree_num_points;       /* Point::tree_num_points: tree roots, 0 elsewhere (cc.cpp:671,822) */
    uint32_t* cluster_width;         /* Point::cluster_width: tree roots, 0 elsewhere (cc.cpp:666,819) */
    uint32_t* number_of_child_points;/* Point::child_points.size() (cc.cpp:663) */
    int32_t* number_of_visited_neighbors; /* Point::number_of_visited_neighbors (cc.cpp:725) */
    uint8_t* belongs_to_finished_cluster; /* Point::belongs_to_finished_cluster: set on tree roots only (cc.cpp:933) */
    int64_t* tree_parent_global_column;   /* the point whose child_points list holds this point (cc.cpp:663); -1 = tree root / no tree */
    int32_t* tree_parent_row;
} cc_column_view;

typedef struct cc_engine cc_engine;

/* Create an engine on HIP device `device` for `num_streams` sensor streams with `num_rows` lasers.
 * Equivalent of constructing num_streams ContinuousClustering objects, setConfiguration(cfg) and
 * reset(num_rows) (cc.cpp:9-81). Fails with CC_ERR_NO_DEVICE when no GPU is present. */
int cc_engine_create(cc_engine** out, int device, int num_streams, int num_rows, const cc_config* cfg);
void cc_engine_destroy(cc_engine* e);

/* setConfiguration (cc.cpp:66-81): stores the configuration, recomputes max_distance^2 and raises
 * reset_required on every stream when is_single_threaded / sensor_is_clockwise / num_columns change. */
int cc_engine_set_config(cc_engine* e, const cc_config* cfg);
/* reset(num_rows) (cc.cpp:11-64) for all streams. */
int cc_engine_reset(cc_engine* e, int num_rows);
/* setTransformRobotFrameFromSensorFrame (cc.cpp:626-631). tf = 3x4 row-major [R|t] in double;
 * stream = -1 sets it for all streams. */
int cc_engine_set_robot_from_sensor(cc_engine* e, int stream, const double tf[12]);

/* addFiring x n for ONE stream from host buffers (cc.cpp:88-93 -> the whole call tree of SURVEY 3.1):
 *   xyz        n * num_rows * 3 floats  (RawPoint::x,y,z in the sensor frame; NaN x = no return)
 *   intensity  n * num_rows bytes       (RawPoint::intensity)
 *   poses      n * 12 doubles           (odom_from_sensor as 3x4 row-major [R|t])
 * Returns when all n firings have been inserted and every column they finish has been segmented,
 * associated, checked and published; events are queued in reference order.
 * Columns published during a call stay readable (cc_engine_read_columns) until the next call when
 * n <= 2 * num_columns; longer calls are split internally and may clear (cc.cpp:1091) what their first
 * part published. Calls with n <= 8 take a low-latency path: one captured hipGraph launch — on sensors of <= 64 rows three kernels that read the
 * firings from pinned staging and write state and events back into pinned memory, no copy node, the host spins on a sequence number
 * (one firing per call: ~48 us p50); otherwise H2D of the firings, every kernel, D2H of state and events and one synchronisation. */
int cc_engine_add_firings(cc_engine* e, int stream, int64_t n, const float* xyz, const uint8_t* intensity,
                          const double* poses);

/* Throughput entry: advance ALL streams by n firings each from DEVICE-resident buffers
 *   d_xyz        [num_streams][n][num_rows][3] float
 *   d_intensity  [num_streams][n][num_rows]    uint8
 *   d_poses      [num_streams][n][12]          double
 * Launches on the engine's HIP streams and returns without synchronising. The buffers must be complete when the call is made
 * (the kernels that read them run on internal streams), unless their producer was enqueued on cc_engine_hip_stream(e) AND the
 * option "input_on_engine_stream" is set: then the engine orders its reads after that work.
 * LIFETIME OF THE INPUTS: the call is asynchronous and so are its reads. The three buffers must stay allocated and UNCHANGED until the
 * engine has released them: cc_engine_inputs_released() (below; it never waits) reports the last call whose buffers will not be read again,
 * and cc_engine_sync() — or any call that implies it: cc_engine_stream_state, cc_engine_drain_events, cc_engine_read_columns,
 * cc_engine_reset ... — releases everything submitted before it. A caller that streams batches in without synchronising keeps its buffers
 * in a ring and, before it overwrites a buffer, asks cc_engine_inputs_released whether that buffer's call is through (in steady state a
 * call's inputs are read for the last time while the second or third call after it runs, later after a lazy-gate miss: do not count calls,
 * ask). Re-using a buffer earlier is a data race the engine cannot see by itself — the symptom is "reset_required" on a healthy stream —
 * unless the option "check_input_lifetime" is set (a debugging aid: it checksums every call's inputs at submission and at release and
 * fails the release with CC_ERR_INVALID_ARGUMENT, naming the call, when they differ; 2: released buffers are also overwritten with 0xFF). */
int cc_engine_add_firings_device(cc_engine* e, int64_t n, const float* d_xyz, const uint8_t* d_intensity,
                                 const double* d_poses);
/* Which input buffers of cc_engine_add_firings_device may be re-used. Calls are numbered 1, 2, ... per engine (the numbers run on over
 * cc_engine_reset). *released_call = the last call whose d_xyz / d_intensity / d_poses the engine will not read again (0: none yet),
 * *submitted_calls = calls made so far; either pointer may be NULL. Never waits and launches nothing (it queries events recorded behind
 * each call's last kernel chain); the answer only grows. A call whose chains the engine still holds back (deferred tail, lazy gate) or
 * that went through the non-pipelined path is reported once a later call or a synchronisation has put it through. With the option
 * "check_input_lifetime" this is also where a buffer that changed while the engine owned it is reported (CC_ERR_INVALID_ARGUMENT). */
int cc_engine_inputs_released(cc_engine* e, uint64_t* released_call, uint64_t* submitted_calls);
/* Block until everything launched so far has finished. */
int cc_engine_sync(cc_engine* e);
/* The hipStream_t all engine work is enqueued on (for hipEvent timing by the caller). */
void* cc_engine_hip_stream(cc_engine* e);

/* Enable / disable event recording (default on for 1 stream, off for >1: throughput mode). */
int cc_engine_record_events(cc_engine* e, int enable);
/* Move up to `capacity` queued events of `stream` into `out`; *n = number written. Implies sync. */
int cc_engine_drain_events(cc_engine* e, int stream, cc_event* out, int64_t capacity, int64_t* n);

/* Tree links made since the last drain (cc.cpp:693-694, associatePointTreeToPointTree), each as four int64:
 * (global column, row) of the one tree root, (global column, row) of the other. Only recorded while events are (option
 * "mirror_fields"); a host mirror rebuilds Point::associated_trees from them to walk a finished cluster's trees in the
 * reference's order (cc.cpp:851-910). capacity = 0: *n = number of queued links, nothing is removed. Implies sync. */
int cc_engine_drain_links(cc_engine* e, int stream, int64_t* out, int64_t capacity, int64_t* n);

/* Number of queued events of `stream`. Implies sync. */
int cc_engine_pending_events(cc_engine* e, int stream, int64_t* n);

int cc_engine_stream_state(cc_engine* e, int stream, cc_stream_state* out);
/* Copy the columns [from, to] (global indices, inclusive, to - from < ring_buffer_max_columns) of `stream` to host. */
int cc_engine_read_columns(cc_engine* e, int stream, int64_t from, int64_t to, const cc_column_view* view);
/* The same for up to 8 ranges [from[i], to[i]] at once (the ranges' columns one after the other in the view's arrays, sum(to - from + 1) columns in
 * all): what a front-end that keeps a mirror of range_image_ reads per call — the newest column (ground view) and the oldest ones (published) lie a
 * lag of ~100 columns apart. One kernel launch and one copy whatever the number of ranges; none at all when the last call on the stream was a small
 * one (< 64 firings) and the ranges only name columns it segmented or published (at most 8: option "mirror_views", cc_engine_view_counters). */
int cc_engine_read_column_ranges(cc_engine* e, int stream, int n_ranges, const int64_t* from, const int64_t* to, const cc_column_view* view);

/* Device pointers of the two per-cell OUTPUT planes of a stream's ring buffer (ground label u8,
 * cluster id u32), indexed [local_column * num_rows + row] — what the label-compare kernels and
 * bench checksums read without leaving HBM. */
int cc_engine_output_planes(cc_engine* e, int stream, const uint8_t** d_ground_label, const uint32_t** d_cluster_id);

/* Cluster hand-off: the member points of n finished clusters of `stream`, gathered and compacted on the device — the point
 * collection of collectPointsForCusterAndPublish (cc.cpp:985-1033; the reference walks the point trees, this returns the same set
 * ordered by global column, then row). Describe every cluster by its CC_EV_CLUSTER event: cluster_ids[i] = event.c,
 * col_from[i] = event.a, col_to[i] = event.b, n_points[i] = event.d. Cluster i receives elements
 * [sum(n_points[0..i)), sum(n_points[0..i])) of h_gcol / h_row (host buffers of sum(n_points) elements). Call it before the
 * cluster's columns leave the ring (i.e. right after the call that finished it). CC_ERR_INVALID_ARGUMENT if a descriptor does
 * not match what the engine holds. */
int cc_engine_gather_cluster_points(cc_engine* e, int stream, int64_t n, const uint32_t* cluster_ids, const int64_t* col_from,
                                    const int64_t* col_to, const uint32_t* n_points, int64_t* h_gcol, int32_t* h_row);

/* Engine tuning / test hooks: one option per line, `name` (default) meaning. Values out of range are clamped. Environment variables that override
 * options (CC_ASSOC_ROUNDS, CC_DEFER_TAIL, ... as used by the A/B tools) are only read when CC_ENABLE_ENV_OPTS=1 is set. Every setting gives the
 * same results (the parity tests run over them, tests/test_gpu_stress.py walks random combinations); they only move work between kernels and streams.
 *
 *  -- pipeline shape of cc_engine_add_firings_device ------------------------------------------------------------------------------------------
 *  "pipeline"                (2)   0: the kernel chains of a batch back to back on one HIP stream; 1: consecutive batches overlap on three chains
 *                                  (insertion | table, segmentation, window scan | association); 2: the window scan on a fourth chain
 *  "sub_batch"               (0)   firings per pipelined sub-batch of one call; 0: the whole call is one batch
 *  "limit_columns"                 columns one launch may emit per stream before it hands back to the host (continuation passes)
 *  "publish_off_chain"       (1)   pipelined mode: k_publish on a stream of its own instead of at the end of the association chain
 *  "table_on_insert_chain"   (1)   pipelined mode: k_table at the end of the insertion chain; 0: at the head of the segmentation chain; 2: own stream
 *  "ego_on_insert_chain"     (0)   1: k_ego next to k_table instead of in front of k_seg_pre
 *  "ego_off_chain"           (0)   experiment: pipelined mode with the fused front half: k_ego of a batch on the preparation stream, beside the previous
 *                                  batch's insertion, instead of in front of its own insertion (32 streams - 8 .. - 12 %: the event costs more than the kernel)
 *  "input_on_engine_stream"  (0)   1: the caller's device buffers are produced by work enqueued on cc_engine_hip_stream(e) (cc_kitti_convert_frames)
 *  "defer_tail_max_streams"  (96)  launches of at most that many streams leave the chains behind a batch's insertion gate to the NEXT call, which
 *                                  launches them behind its own insertion; every call that reads, synchronises or resets flushes them first; 0: never
 *  "lazy_gate"               (40)  launches of at most that many streams enqueue their insertion before the host has read the PREVIOUS batch's
 *                                  insertion counters; the kernels check them on the device and return if that batch still needs the serial insertion
 *                                  kernels (the host then launches those and this insertion again: cc_engine_gate_counters). Two misses in a row
 *                                  switch it off, eight clean batches or cc_engine_reset switch it on again; 0: never
 *  "lazy_gate_from"          (80)  ... and launches of at least that many streams (96 streams + 8 .. 10 %, 256 + 2 .. 5 %, nothing at 41 .. 79); 0: none
 *  -- insertion ----------------------------------------------------------------------------------------------------------------------------------
 *  "parallel_insert"         (1)   the head of every batch of >= 64 firings that has the single-column firing shape is inserted by a block-parallel
 *                                  kernel (k_insert_par up to 64 rows, k_insert_multi above), the serial kernel continues behind it; 0: serial only
 *  "skip_idle_fallbacks"     (1)   pipelined mode: the host waits for the block-parallel insertion kernel of a batch and launches the other insertion
 *                                  kernels only if some stream's batch was not taken completely; 0: always launch them
 *  "fuse_front"              (1)   k_insert_par also does the per-cell part of the ground segmentation of the columns it fills and closes batches it
 *                                  took completely as fused (k_table / k_seg_pre only for streams that need them); 0: the unfused chain
 *  "insert_wide_max_streams" (160) launches of at most that many streams run k_insert_par with 16 wavefronts per block ...
 *  "insert_split_blocks"     (0)   ... and deal a stream's firings to that many blocks; 0: 8 up to 24 streams, 6 up to 32, 4 up to 40, 3 up to 64,
 *                                  2 up to 96, else 1
 *  "insert_narrow_blocks"    (0)   experiment: that many 4-wavefront blocks per stream above insert_wide_max_streams
 *  "insert_lds_pad"          (0)   experiment: KB of unused dynamic LDS that keep a second insertion block off a compute unit
 *  -- segmentation, window scan ------------------------------------------------------------------------------------------------------------------
 *  "seg_small_max"           (63)  calls of at most that many firings on a sensor of <= 64 rows segment with k_seg_small (rows as lanes)
 *  "scan_packed"                   1: the packed window scan k_scan2 (default above 192 streams per launch and at 128 rows); 0: k_scan
 *  "scan_split"              (2)   throughput mode: a point of the packed scan that is still scanning after 6 visits (it found no neighbour:
 *                                  vegetation, spray) is handed to k_scan2_long, which runs such points with every lane busy, and k_scan2_epi
 *                                  finishes its column. 1: always; 0: one pass; 2: while such scans are a large part of the work (every 32nd batch
 *                                  is scanned this way and counted: on above 40 visits of k_scan2_long per column of 64 rows, off again below 20)
 *  "scan_cap"                (6)   ... the number of visits after which a point is handed over
 *  "scan_store_fin"          (-1)  a point's contribution to its tree's finished_at, for the serial association kernels: 1 the window scan stores it per
 *                                  cell (8 B), 0 they recompute it, -1 per launch: stored where they are expected to associate (assoc_batch 0, pinned
 *                                  rounds, a stop of k_assocb within the cool-down), not behind an undisturbed k_assocb, which gets the value packed
 *  "scan_long_records"       (8192) room of a stream's list of such points per batch (a lane that finds it full finishes its scan in place)
 *  -- association -----------------------------------------------------------------------------------------------------------------------------------
 *  "assoc_batch"             (1)   the batch-parallel kernel k_assocb runs in front of the serial one and takes every group of columns that cannot
 *                                  differ from the sequential semantics (cc_engine_batch_counters); 0: serial kernels only
 *  "assoc_rounds"            (0)   1..8 (batch-parallel, serial) kernel pairs per batch; 0: adaptive — one pair, three for "assoc_cooldown" batches
 *                                  after k_assocb had to stop; all but the last serial launch only take the group k_assocb stopped in front of
 *  "assoc_cooldown"          (4)   see assoc_rounds
 *  "assoc_sweep_blocks"      (2)   blocks of the serial kernel while it is only the safety net behind k_assocb (they sweep over all streams)
 *  "assoc_waves"             (0)   wavefronts per stream of the serial kernel. 0: k_assoc3 (resolve / records / apply) plus its links wavefront while a
 *                                  launch has <= 256 streams; 3 / 4: without / with the links wavefront; 1: the one-wavefront kernel (also what
 *                                  cluster_point_trees_every_nth_column != 1 uses)
 *  "lds_tree_limit"                unfinished point trees per stream kept in LDS before the stream continues in the global-memory kernel, 1..256
 *  "mirror_fields"                 1 (default while events are recorded): also produce what only a host mirror of range_image_ shows
 *                                  (number_of_visited_neighbors, per-tree values of finished trees, the tree-link log); 0 in throughput mode
 *  -- small calls of cc_engine_add_firings (per-column latency path) -----------------------------------------------------------------------------
 *  "graphs"                  (1)   0: never use the captured-hipGraph path of small calls
 *  "small_front"             (1)   a call of < 64 firings on ONE stream outside the pipeline runs k_small_front / k_small_tail, results mirrored into
 *                                  pinned host memory
 *  "small_all"               (1)   ... as ONE launch, k_small_all, on a 64-row engine; the host launches the serial fall-back behind it when asked
 *  "small_direct"            (1)   calls of 9 .. 63 firings are one direct launch of k_small_all; up to 8 firings replay a captured one-node graph
 *  "mirror_views"            (1)   a small call (k_small_all / the resident kernel) writes the host views of the columns its events name — what it
 *                                  segmented, what it published, if at most 8 — into pinned memory with its results: cc_engine_read_columns for such
 *                                  columns needs no kernel, no copy and no synchronisation (cc_engine_view_counters)
 *  "resident"                (0)   1: calls of 1 .. 63 firings on a 1-stream 64-row engine are handed to a RESIDENT kernel (k_resident: one block that
 *                                  stays on a compute unit) through a doorbell in pinned memory — no dispatch per call. Started by the first such
 *                                  call; read-only queries (cc_engine_read_columns, cc_engine_gather_cluster_points, the event drains) run beside
 *                                  it on another stream; everything else (reset, set_config, set_option, larger calls, destroy ...) stops it first
 *  "resident_idle_ms"        (20)  the kernel's watchdog: it leaves by itself after that long without a call (a host thread that went away must
 *                                  not pin a compute unit); the next call launches it again
 *  "prewarm_small_graphs"          one-shot action (value k in 1..8): size the grow-only buffers of small calls and capture the graphs of calls of
 *                                  1..k firings now, without launching anything
 *  "forget_inclination_table"      one-shot action: clear the ground-segmentation inclination table too (the only state cc_engine_reset keeps, like
 *                                  the reference's reset(), cc.cpp:46); used by the drop-in class after its warm-up
 *  -- debugging --------------------------------------------------------------------------------------------------------------------------------------
 *  "debug_flags"             (0)   experiment switches
 *  "debug_no_assoc_fallback" (0)   1: do not launch the serial kernels behind k_assocb (tools only: shows what k_assocb alone covers)
 *  "timing_every"            (1)   with cc_engine_enable_timing on: the ten HIP events bracket every n-th batch only; cc_engine_kernel_times returns
 *                                  the sampled sums scaled to all batches (average launch duration x launches). The events cost 15 - 25 us of the
 *                                  association chain per batch: 3.5 - 5 % of a step at 32 - 64 streams, 1 % at 256 (profiles/r06_ab_timing.txt)
 *  "check_input_lifetime"    (0)   1: checksum the inputs of every cc_engine_add_firings_device call at submission and again at release
 *                                  (cc_engine_inputs_released, cc_engine_sync ...): a caller that re-used a buffer too early gets
 *                                  CC_ERR_INVALID_ARGUMENT naming the call instead of a silent race; 2: released buffers are also filled with 0xFF */
int cc_engine_set_option(cc_engine* e, const char* name, int64_t value);

/* Per-kernel timing with HIP events recorded on the engine's streams around the kernels of every batch (or every n-th: option
 * "timing_every"): prep, insert+table, segment, scan, assoc_lds, assoc_global, publish. enable resets the accumulators;
 * cc_engine_kernel_times returns the accumulated milliseconds and the number of batches they stand for. */
int cc_engine_enable_timing(cc_engine* e, int enable);
int cc_engine_kernel_times(cc_engine* e, double ms[7], uint64_t* launches);
/* Sums over all streams (any pointer may be NULL). Implies sync. */
int cc_engine_totals(cc_engine* e, uint64_t* cells_published, uint64_t* clusters_finished, uint64_t* firings_consumed,
                     uint64_t* serial_columns);
/* Sums over all streams: columns whose association + finished-cluster check (continuous_clustering.cpp:773-974) ran in the batch-parallel
 * kernel (groups of up to 64 columns at once), and how often that kernel handed the rest of a batch to the exact serial kernel because a
 * group could have differed from the reference's sequential semantics (attach to a finished tree, one-rotation limits, look-back past the
 * first unpublished column, ...). Option "assoc_batch" = 0 takes the batch-parallel kernel out. Implies sync.
 * bail_reasons[i] (may be NULL) counts them by reason: 1 more unfinished trees than the kernel's lanes, 2 link overflow, 3 a tree / cluster
 * could reach the one-rotation limits (cc.cpp:657, 913-924), 4 a parent chain ends in a finished tree (cc.cpp:658), 5 a tree receives a point
 * after its cluster finished, 6 a candidate from a column older than the first unpublished one (cc.cpp:762-763); bail_reasons[7] is not a
 * reason: the number of small cc_engine_add_firings calls (one launch, k_small_all) whose serial fall-back kernel the host had to launch behind it. */
int cc_engine_batch_counters(cc_engine* e, uint64_t* batch_columns, uint64_t* batch_bails, uint64_t bail_reasons[8]);
/* The resident single-stream kernel (option "resident"): launches of k_resident so far, calls it has answered, and whether it sits on the
 * engine's stream right now (it leaves by itself after "resident_idle_ms" without a call, and whenever a call needs the host). The number of
 * calls of a launch that is still running is as of its last exit report (0 until then). Any pointer may be NULL. Never waits. */
int cc_engine_resident_counters(cc_engine* e, uint64_t* launches, uint64_t* calls, int* running);
/* cc_engine_read_columns calls served from the views a small call mirrored with its results (option "mirror_views": the columns the call
 * segmented and published, at most 8; requests that name only such columns and do not ask for number_of_child_points) / by the view kernel. */
int cc_engine_view_counters(cc_engine* e, uint64_t* served_from_mirror, uint64_t* served_by_kernel);

/* The insertion gate of the pipelined mode (option "lazy_gate"): how many batches had their insertion enqueued before the host had read the
 * previous batch's insertion counters, and how many of those were launched a second time because the previous batch turned out to need the
 * serial insertion kernels first (the kernels of the first launch return at once in that case). No sync. */
int cc_engine_gate_counters(cc_engine* e, uint64_t* lazy_batches, uint64_t* lazy_redone);

/* ---- label compare (src/evaluation/kitti_evaluation.cpp) ---------------------------------------------------------------- */
/* EvaluationResultForFrame, kitti_evaluation.hpp:38-49 */
typedef struct cc_eval_frame_result
{
    double tp, fn, fp, tn;
    double over_segmentation_entropy;
    double under_segmentation_entropy;
} cc_eval_frame_result;

/* KittiEvaluation::evaluate for one frame (kitti_evaluation.cpp:29-146): ground confusion counts over the points whose semantic
 * label is not "unlabeled", OSE / USE between euclidean ground-truth labels and detection ids. Arrays have n entries (one per
 * point of the frame): semantic label (SemanticKITTI numeric id), euclidean-clustering label, is_ground flag, detection id.
 * cc_eval_frame takes host arrays, cc_eval_frame_device device arrays (current HIP device). */
int cc_eval_frame(int device, int64_t n, const uint16_t* semantic, const uint32_t* euclid, const uint8_t* is_ground,
                  const uint32_t* detection, cc_eval_frame_result* out);
int cc_eval_frame_device(int64_t n, const uint16_t* d_semantic, const uint32_t* d_euclid, const uint8_t* d_is_ground,
                         const uint32_t* d_detection, cc_eval_frame_result* out);
/* calculateMeanAndStdDev (kitti_evaluation.cpp:277-293) */
void cc_eval_mean_std(const double* data, int64_t n, double* mean, double* std_dev);
/* the six {mean, sigma} pairs of generateEvaluationResults (kitti_evaluation.cpp:187-208) over n frames: recall, precision, F1,
 * accuracy (as fractions), USE, OSE */
void cc_eval_summarize(const cc_eval_frame_result* frames, int64_t n, double out[12]);

/* The one collective of the system: every rank's per-frame records — 8 doubles each: sequence, frame, then the six values of
 * EvaluationResultForFrame (kitti_evaluation.hpp:38-49) — to every rank, as ONE ncclAllGather of fixed-size padded blocks over RCCL / xGMI.
 * nccl_comm: the caller's ncclComm_t (all ranks call with the same `capacity`, the largest number of records any rank holds); records: host,
 * n x 8; out: host, world x capacity x 8 (rank r's records first in its block); counts[r] = number of records of rank r. RCCL is looked up in
 * the process at run time (no link-time dependency of this library). The reference has no multi-process mode: a single process evaluates all
 * sequences one after the other (kitti_demo.cpp:230-420); with the streams sharded over GPUs this gather restores its view of all records. */
int cc_eval_gather_records(void* nccl_comm, int world, int device, const double* records, int64_t n, int64_t capacity, double* out,
                           int64_t* counts);

/* KittiEvaluation::generateEuclideanClusteringLabels (kitti_evaluation.cpp:224-275, what gt_label_generator_tool.cpp:50-70 writes to
 * labels_euclidean_clustering/): connected components of {squared distance < 1 m^2, same semantic and instance label} over one frame
 * (PCL ConditionalEuclideanClustering, tolerance 1.0, 10..300000 points), numbered 1, 2, ... in the order their first point appears; 0 for
 * dropped components and for points of the ground / unlabeled classes. points = n x 4 floats (x, y, z, i), host arrays. */
int cc_eval_generate_euclidean_labels(int device, int64_t n, const float* points, const uint16_t* semantic, const uint16_t* instance,
                                      uint16_t* out_labels);

/* Human-readable text of the last failing call on this engine (never NULL). */
const char* cc_engine_last_error(cc_engine* e);
/* Library / build identification, e.g. "continuous_clustering_amd 0.1 gfx950". */
const char* cc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CC_HIP_H */
