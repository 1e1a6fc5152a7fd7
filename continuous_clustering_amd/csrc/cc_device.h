// cc_device.h — device-side data layout of one cc_engine (gfx950).
//
// Layout in HBM (DESIGN.md "Data layout"): every per-cell field of the reference's 232-byte AoS
// `Point` (continuous_clustering.hpp:126-161) that the algorithm owns is its own plane
// plane[stream][local_column * num_rows + row] — column-major like the reference's range_image_
// (continuous_clustering.cpp:181) so that a 64-lane wavefront whose lanes are the rows of one
// column reads and writes 64 consecutive elements. Per-tree state (the fields the reference keeps
// in the root Point) lives in planes indexed by the root cell. Per-stream scalars live in
// StreamState.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#include "../../include/cc_hip.h"

namespace ccd
{

constexpr int WAVE = 64;
constexpr int MAX_ROWS_PER_LANE = 2; // num_rows <= 128
constexpr int LINK_SLOTS = 4;        // link candidates recorded per point by the static window scan
constexpr int WIN_COLS = 32;         // columns of tree-slot ids kept in LDS by the association kernel
constexpr int PP_SKIP = 0x7fffffff;
#ifndef CC_INS_WIN
#define CC_INS_WIN 64
#endif
constexpr int INS_WIN = CC_INS_WIN;          // columns of `distance` kept in LDS by the insertion kernel
constexpr int SG_NAN = 1, SG_FOG = 2, SG_EGO = 4, SG_INCL_IGNORE = 8, SG_TOO_CLOSE = 16, SG_PENDING = 32; // (k_seg_pre / k_insert_par -> k_seg_scan)
#ifndef CC_TREE_SLOTS
#define CC_TREE_SLOTS 256
#endif
constexpr int TREE_SLOTS = CC_TREE_SLOTS;      // unfinished point trees per stream kept in LDS (more -> global-memory kernel)
// k_table / k_seg_pre: the columns a batch segments are cut into TABLE_WAVES ranges of SEGPRE_BLOCKS / TABLE_WAVES chunks (cc_kernels.h)
#ifndef CC_TABLE_WAVES
#define CC_TABLE_WAVES 8
#endif
constexpr int TABLE_WAVES = CC_TABLE_WAVES;
#ifndef CC_SEGPRE_BLOCKS
#define CC_SEGPRE_BLOCKS 256
#endif
constexpr int SEGPRE_BLOCKS = CC_SEGPRE_BLOCKS; // chunks per stream and batch
static_assert(SEGPRE_BLOCKS % TABLE_WAVES == 0, "chunks are distributed evenly over the table wavefronts");
constexpr int BATCH_SLOTS = 4;                  // batch descriptors in flight (StreamState::batch)

// Scalar state of one sensor stream: the srig_*/sgps_*/sc_* members of the reference class
// (continuous_clustering.hpp:244-275) plus engine bookkeeping.
struct StreamState
{
    // continuous range image generation (srig)
    int64_t prev_rearmost;   // srig_previous_global_column_index_of_rearmost_laser
    int64_t prev_foremost;   // srig_previous_global_column_index_of_foremost_laser
    int64_t first_unfinished; // srig_first_unfinished_global_column_index
    int64_t ring_start;      // ring_buffer_start_global_column_index
    int64_t ring_end;        // ring_buffer_end_global_column_index
    int64_t first_column;    // first column ever handed to segmentation since reset (-1 = none)
    int64_t clear_done;      // columns below are physically cleared (clearing lags ring_start by one batch so that the
                             // host can still read what a batch published, cc.cpp:1087-1091)
    int32_t reset_required;
    int32_t has_robot_tf;
    double robot_from_sensor[12]; // sgps_ego_robot_frame_from_sensor_frame_
    // continuous clustering (sc)
    int64_t first_unpublished; // sc_first_unpublished_global_column_index
    uint64_t cluster_counter;  // sc_cluster_counter_
    int32_t n_unfinished;      // size of sc_unfinished_point_trees_
    int32_t pad0;
    int64_t min_required;      // min root column over the unfinished trees (valid when n_unfinished > 0)
    double finish_lower_bound; // lower bound of min over unfinished clusters of max finished_at
    double last_round_min_az;  // column-min azimuth of the previous tree-combination round (the BFS "visited stamp")
    // batch bookkeeping
    int64_t clear_allowed; // ring_start when the current host call began: clearing never passes what the host has seen
    // per-batch hand-off from the insertion chain to the segmentation / association chain; four slots because up to three
    // batches are in flight: b + 2 being inserted, b + 1 segmented / scanned, b associated (three HIP streams)
    struct BatchDesc
    {
        int64_t seg_begin; // columns [seg_begin, seg_end) were emitted by the insertion kernel in this batch
        int64_t seg_end;
        int64_t acp_next;  // next column the association kernels process
        int64_t pub_begin; // columns [pub_begin, pub_end) were published while this batch was associated
        int64_t pub_end;
        int64_t mode;      // assoc_mode as of the start of the batch's segmentation chain (k_table): decides whether k_scan stages the
                           // batch for the LDS association kernels; the global-memory kernel takes the batch if either this or the
                           // current assoc_mode is non-zero
        int64_t fused;     // 1: k_insert_par took the whole batch AND did the per-cell part of its segmentation (staging planes, table carries):
                           // k_table / k_seg_pre have nothing to do for this stream
    } batch[4];
    int32_t assoc_mode; // 0: tree state in LDS (k_assocb / k_assoc3 / k_assoc_lds), 1: tree state in global memory (k_associate); the global kernel
                        // hands a stream back once its unfinished trees fit the LDS pool comfortably again
    int32_t pad1;
    int64_t cursor;     // firings of the current batch already consumed
    int64_t pre_seg_begin; // first column emitted by k_insert_par in this batch (0 = it emitted none): k_insert2 continues its batch
    uint64_t firings_consumed;
    uint64_t cells_published;
    uint64_t clusters_finished;
    uint64_t exceed_one_rotation; // how often cc.cpp:913-919 fired
    uint64_t serial_columns;      // columns that took the exact serial association path
    uint64_t stamp_alias_rounds;  // rounds whose min azimuth equalled the previous round's
    uint64_t batch_columns;       // columns associated by the batch-parallel kernel (k_assocb)
    uint64_t batch_bails;         // launches of k_assocb that handed the rest of their batch to the serial kernel
    uint64_t batch_bail_reason[8]; // ... by reason (cc_assocb.h AB_BAIL_*)
    int64_t par_clear_done;       // k_insert_par over several blocks: the clearing limit the first block to arrive fixed for all of them (-1 since
                                  // k_begin_batch; becomes clear_done in k_insert_par_fin: the other blocks must not see clear_done change),
    int32_t par_bad;              // ... the first firing whose returns left its column (INT_MAX: none; k_begin_batch resets it),
    int32_t par_upto;             // ... and the firing the run ends at by the batch-wide conditions (the same in every block)
    int64_t serial_until;         // set by k_assocb when it stops in front of a group: a LIMITED launch of the serial kernel stops there
    // errors raised inside kernels
    uint64_t dbg[16];             // section cycle counters (CC_PROFILE_SECTIONS builds only): 0-7 insertion, 8-15 association
    int32_t error;
    int32_t n_events;
    int64_t error_a;
    int64_t error_b;
    int32_t n_links;     // entries of the link log written in the current call
    int32_t pad2;
    int64_t overrun_col; // lowest column found stale by the segmentation (CC_ERR_RING_OVERRUN), INT64_MAX = none
};

// All planes of an engine. Index of a cell inside a plane: stream * cells_per_stream + lcol * num_rows + row.
struct Planes
{
    // geometry written by the insertion kernel (x, y, z live in sc_rec below: one 16-byte record per cell)
    float* dist;
    float* incl;
    float* incaz;     // increasing azimuth angle of the return (cc.cpp:146-148). Its continuous azimuth angle (cc.cpp:184-186) is
                      // 2 pi * rotation + incaz, where the rotation is that of the cell's column — or one less when the sign bit is set
                      // (a return moved on to the first column of the next rotation, cc.cpp:188-202)
    uint16_t* gtag;   // which pass over the ring filled the cell (Point::global_column_index in two bytes; 0 = cleared): cc_kernels.h cell_tag
    uint32_t* src;    // sequence number of the firing that filled the cell (low 32 bits)
    uint8_t* inten;
    // per column [stream][lcol]
    int32_t* trig;    // batch-relative index of the firing that finished the column (its pose is the job's pose)
    int64_t* colg;    // global column index of a segmented column
    double* colminaz; // minimum continuous azimuth over the rows of the column
    // ground segmentation output
    uint8_t* ground;
    uint8_t* debug;
    uint8_t* ignored;
    // clustering
    int32_t* root;    // tree root as cell index (lcol * num_rows + row), -1 = none
    uint32_t* id;     // OUTPUT plane: cluster id of published cells
    // per-tree planes, indexed by the root cell
    double* t_fin;    // finished_at_continuous_azimuth_angle
    uint32_t* t_width; // cluster_width
    uint32_t* t_pts;  // tree_num_points
    int32_t* t_uf;    // union-find parent (cell index of another tree root)
    uint32_t* t_cid;  // id of the finished cluster the tree belongs to (0 = none / too small)
    int32_t* t_pos;   // position in the unfinished list
    uint8_t* t_finished; // belongs_to_finished_cluster
    // per-stream pools [stream][tree_capacity]
    int32_t* ulist;   // sc_unfinished_point_trees_ in creation order
    int32_t* ucomp;   // list position of the representative of each listed tree (scratch of the finish check)
    unsigned long long* agg_fin; // per cluster (indexed by list position of its representative)
    long long* agg_min;
    long long* agg_max;
    uint32_t* agg_pts;
    uint32_t* agg_cid;
    int32_t* agg_first;
    uint8_t* agg_flag;
    cc_event* events; // [stream][event_capacity]
    // staging written by the static window scan (k_scan), consumed by k_assoc_lds
    // staging written by k_prep (per input point of the batch: [stream][firing][row]), consumed by k_insert2
    float* pp_x;
    float* pp_y;
    float* pp_z;       // point in the odom frame
    float* pp_dist;
    float* pp_incl;
    float* pp_incaz;   // increasing azimuth angle (cc.cpp:146-148)
    int32_t* pp_cir;   // column index within the rotation (cc.cpp:151); PP_SKIP = no return
    // staging written by k_seg_pre, consumed by k_seg_scan
    float* sg_x2;       // ||xy|| of the point relative to the sensor (to2dInAzimuthPlane(...).x, cc.hpp:229-232)
    float* sg_uz;       // z of the point relative to the sensor
    uint8_t* sg_flags;  // SG_* bits
    float* sg_w;        // the column's own inclination step / the distance / the inclination below (cc_kernels.h: seg_pre_cells)
    // one 16-byte record per cell: {x, y, z, inclination} of the return in the odom frame, written by the insertion kernels (the only
    // copy of x, y, z); cells without a return get {NaN, NaN, NaN, supplemented inclination} from k_seg_pre. What the window scan
    // reads per visited cell.
    float4* sc_rec;
    // candidates are coded as (columns back << 8) | row
    int16_t* sc_parent;  // first accepted candidate, -1 = none, -2 = point is ignored
    int16_t* sc_term;    // where the point's same-column parent chain ends: >= 256 candidate code (delta << 8 | row) in an earlier
                         // column, 0..255 index of the chain's new root among the column's new roots, -1 point is ignored
    double* col_newfin;  // per column: minimum finished_at over the column's new roots (+inf without any)
    int32_t* col_info;   // per column: new roots | flags << 8 (1: link overflow, 2: any links) | largest delta used << 16
    unsigned* pk_meta;   // the active points of a column packed in row order (entry j of local column lc at lc * rows + j): term | row << 16 | links << 23 | root << 26
    double* pk_fin;      // ... their finished_at contributions
    unsigned long long* pk_lk; // ... their link candidate codes (written where the point has any)
    uint16_t* col_act;   // per column: active points (cells the window scan ran for) of rows 0 - 63 | of rows 64 - 127 << 8 (k_assocb packs them into lanes)
    uint8_t* sc_nlinks;  // accepted candidates after the first one, 255 = more than LINK_SLOTS
    unsigned long long* sc_links; // LINK_SLOTS x 16-bit candidate codes packed into one word per cell
    double* sc_fin;      // continuous azimuth + max angle diff of the point (its contribution to finished_at): written by the window scan only for
                         // batches whose launch has Geometry::scan_stores_fin set (the serial kernels are expected to associate); otherwise the
                         // serial kernels recompute it (cc_k_base.h: cell_fin) and the batch-parallel kernel gets it packed (pk_fin)
    uint16_t* sc_visits; // Point::number_of_visited_neighbors (cc.cpp:725), only with Geometry::mirror_fields
    int2* link_log;      // [stream][link_capacity] (root cell, root cell) of every tree link made in the current call (cc.cpp:693-694), only
                         // with Geometry::mirror_fields: the host rebuilds Point::associated_trees from it
    // long scans of the packed window scan (cc_k_scan.h: k_scan2<.., SPLIT> -> k_scan2_long -> k_scan2_epi)
    void* sl_rec;     // [stream][SL_CAP] ScanLongRec: points that were still scanning after SCAN_CAP visits
    int32_t* sl_ctl;  // [stream][4] records | next record to hand out | deferred columns | blocks of k_scan2_epi that are through
    int32_t* sl_cols; // [stream][ring_cols] local columns whose epilogue waits for k_scan2_long
    int32_t* par_off; // [stream][IP_MAXF] column offset of every firing of the batch (k_insert_par over several blocks -> k_insert_par_fin)
    float* curtab;    // [stream][num_rows] sc_inclination_angles_between_lasers_ after the last emitted column
    unsigned long long* tab_acc; // [stream][Geometry::tab_tiles][num_rows] (column + 1) << 32 | bits of the last valid inclination step inside a tile, as the
                      // wavefronts of the fused insertion find them (atomic max); k_insert_par / k_insert_par_fin turn them into tabc and wipe them
    // what k_table leaves for k_seg_pre (one set per batch-descriptor slot; the engine passes the slot's pointers):
    float* tabc;      // [stream][Geometry::tab_tiles][num_rows] sc_inclination_angles_between_lasers_ as of the column in front of each tile of 64
                      // columns of the batch (k_table, or k_insert_par for batches it segmented itself)
};

struct Geometry
{
    int32_t num_streams;
    int32_t num_rows;
    int32_t num_columns;
    int32_t ring_cols;       // ring_buffer_max_columns = 10 * num_columns (cc.cpp:17)
    int64_t cells;           // ring_cols * num_rows
    int32_t tree_capacity;
    int32_t event_capacity;
    float az_width;          // srig_azimuth_width_per_column
    float max_distance_squared;
    int32_t record_events;
    int32_t limit_columns;   // a launch stops consuming firings of a stream once it emitted this many columns
    int32_t debug_flags;     // experiment switches (cc_engine_set_option "debug_flags"); 0 in production
    int32_t lds_tree_limit;  // unfinished trees kept in LDS before a stream falls back to the global-memory kernel (<= TREE_SLOTS)
    int32_t mirror_fields;   // also produce the per-point fields only the host mirror of range_image_ shows (visited-neighbour counts, the
                             // parent of live-replayed points, per-tree values of finished trees, the tree-link log)
    int32_t link_capacity;
    int32_t tab_tiles;       // tiles of 64 columns a batch can have: entries of Planes::tabc per stream and batch-descriptor slot
    int32_t scan_cap;        // visits a lane of the packed window scan spends on its point before it hands it to the long-scan list (option "scan_cap")
    int32_t scan_stores_fin; // per LAUNCH (cc_engine.hip sets it in the copy it hands to a batch's window scan and serial association kernels, 0 elsewhere):
                             // 1 = the scan writes Planes::sc_fin and the serial kernels read it, 0 = nobody writes it and they recompute (cell_fin)
    int32_t sl_cap;          // records of a stream's long-scan list the packed window scan may use (<= SL_CAP; option "scan_long_records": tests shrink it)
};

} // namespace ccd
