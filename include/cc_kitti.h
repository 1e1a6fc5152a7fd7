/* cc_kitti.h — C-ABI of the KITTI replay path that sits immediately upstream of insertion.
 *
 * The reference turns one KITTI / SemanticKITTI velodyne frame (an unorganised, ego-motion-corrected cloud in a .bin
 * file) into 2200 pseudo-firings of 64 lasers that it feeds to ContinuousClustering::addFiring
 * (src/tools/kitti_demo.cpp:352-395). The per-point steps of that conversion are KittiLoader methods
 * (src/evaluation/kitti_loader.cpp):
 *
 *     recoverLaserIndices      :48-99    row of every point from the azimuth jumps of the file order
 *     undoEgoMotionCorrection  :177-210  per-point rigid transform picked from a 1-ms bin table
 *     generateRangeImage       :101-175  azimuth binning into 64 x 2200 cells with the shift-if-occupied rule
 *     makePseudoFiringFromRangeImageColumn  kitti_demo.cpp:123-159  one firing per range-image column
 *
 * This library runs them as HIP kernels on gfx950 and leaves the firings in HBM in exactly the layout
 * cc_engine_add_firings_device (cc_hip.h) consumes, so a replayed frame never returns to the host between the .bin
 * upload and the published columns. The small per-frame / per-firing pose arithmetic of the same call sites (slerp between
 * two poses, the bin table) is host code behind the same ABI. There is no CPU variant of the per-point steps.
 *
 * All poses are 3x4 row-major [R|t] doubles (12 values), stamps are nanoseconds. Functions return CC_OK (0) or a CC_ERR_*
 * code of cc_hip.h; cc_kitti_last_error() has the text.
 */
#ifndef CC_KITTI_H
#define CC_KITTI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* KittiLoader::RANGE_IMAGE_HEIGHT / RANGE_IMAGE_WIDTH (kitti_loader.hpp:84-86) */
#define CC_KITTI_ROWS 64
#define CC_KITTI_COLS 2200

typedef struct cc_kitti cc_kitti;

/* which steps one cc_kitti_convert_frames call runs for a frame */
enum
{
    CC_KITTI_RECOVER_ROWS = 1,    /* recoverLaserIndices; without it the rows come from cc_kitti_frame::laser_index */
    CC_KITTI_UNDO_EGO_MOTION = 2, /* undoEgoMotionCorrection with cc_kitti_frame::bin_transforms */
    CC_KITTI_RANGE_IMAGE = 4,     /* generateRangeImage */
    CC_KITTI_SHIFT_OCCUPIED = 8,  /* its shift_cell_if_already_occupied argument (default true in the reference) */
    CC_KITTI_FIRINGS = 16         /* makePseudoFiringFromRangeImageColumn for all 2200 columns (needs RANGE_IMAGE) */
};
#define CC_KITTI_ALL_STAGES 31u

typedef struct cc_kitti_frame
{
    const float* points;        /* n_points x 4 floats (x, y, z, i) — the .bin payload (loadPointCloud, kitti_loader.cpp:12-29); host */
    int64_t n_points;
    const uint8_t* laser_index; /* host, n_points; read when CC_KITTI_RECOVER_ROWS is not set (NULL = all rows 0) */
    uint32_t stages;
    int32_t num_bins;                /* rows of bin_transforms */
    uint64_t rotation_start_stamp;   /* undoEgoMotionCorrection arguments */
    uint64_t rotation_end_stamp;
    const double* bin_transforms;    /* host, num_bins x 12: velodyne_from_velodyne of every 1-ms bin (cc_kitti_bin_transforms) */
    /* CC_KITTI_FIRINGS outputs, DEVICE pointers (NULL = not wanted): */
    float* d_xyz;              /* [2200][64][3] RawPoint::x,y,z of firing c, laser r (NaN = empty cell) */
    uint8_t* d_intensity;      /* [2200][64]    static_cast<uint8_t>(i * 255), 0 for empty cells */
    int32_t* d_original_index; /* [2200][64]    KittiPoint::original_kitti_index, -1 for empty cells */
} cc_kitti_frame;

typedef struct cc_kitti_frame_info
{
    int32_t rows_found;   /* laser_index + 1 at the end of recoverLaserIndices (kitti_loader.cpp:92); 64 for a healthy frame */
    int32_t max_columns;  /* its max_columns statistic (:80); the reference throws when it exceeds 2200 (:96-97) */
    int64_t break_index;  /* first point of the 65th row (:74-76), n_points if there is none; points from here on keep row 0 */
    int64_t skipped;      /* points whose azimuth is NaN (the reference indexes out of bounds for them; they are left out) */
} cc_kitti_frame_info;

/* max_frames: frames one cc_kitti_convert_frames call may carry (each gets a device slot); max_points: per frame.
 * hip_stream: the hipStream_t everything is enqueued on (pass cc_engine_hip_stream(e) to chain with an engine); NULL = own stream. */
int cc_kitti_create(cc_kitti** out, int device, int max_frames, int64_t max_points, void* hip_stream);
void cc_kitti_destroy(cc_kitti* k);
const char* cc_kitti_last_error(void);

/* Upload the frames (slot i = frames[i]) and enqueue their stages. Returns without synchronising. */
int cc_kitti_convert_frames(cc_kitti* k, int n_frames, const cc_kitti_frame* frames);
int cc_kitti_sync(cc_kitti* k);
void* cc_kitti_hip_stream(cc_kitti* k);

/* Results of slot `slot` of the last cc_kitti_convert_frames call (synchronises). Any pointer may be NULL.
 *   h_points       n_points x 4: the points after undoEgoMotionCorrection (or as uploaded)
 *   h_laser_index  n_points
 *   h_cell_source  [64][2200] (row-major like the reference's organized_points, kitti_loader.cpp:165): the original index of the
 *                  point that generateRangeImage leaves in each cell, -1 for an empty cell */
int cc_kitti_frame_result(cc_kitti* k, int slot, cc_kitti_frame_info* info, float* h_points, uint8_t* h_laser_index,
                          int32_t* h_cell_source);

/* ---- the frame scatter of the harness, addColumnAndEvaluateFrameIfCompleted (kitti_demo.cpp:173-224), on the device --------------------
 * For an engine whose streams are each fed exactly num_columns pseudo-firings per frame (kitti_demo.cpp:386-403; cc_kitti_convert_frames
 * writes them straight into the engine's input arrays). A published cell then names its KITTI point through the sequence number of the
 * firing that filled it: frame = sequence / num_columns, column of the frame = sequence % num_columns, point =
 * d_original_index[stream][frame % slots][column][row] (cc_kitti_frame::d_original_index of that frame's conversion; `slots` frames are
 * kept per stream, 4 are enough: a column is published within a rotation of its insertion).
 *   cc_engine_scatter_info   per published column of the ranges [from[i], to[i]] of streams[i] (concatenated in the outputs): the smallest
 *                            and largest frame index among the column's points (INT32_MAX / -1 for a column without points) — what the
 *                            harness needs to find the column in which frame N + 1 starts (:205-209) and its two error cases (:203-206).
 *   cc_engine_scatter_apply  is_ground_point = (ground_point_label == GP_GROUND) and detection_label = id (:214-215) of the points of the
 *                            columns [from, to] into d_is_ground / d_detection, laid out [stream][frame % slots][max_points]; asynchronous
 *                            on cc_engine_hip_stream(e). cc_eval_frame_device (cc_hip.h) evaluates a frame from those arrays.
 * Both are declared here because they belong to the replay harness; they live in the engine (include cc_hip.h first). */
struct cc_engine;
int cc_engine_scatter_info(struct cc_engine* e, int n, const int32_t* streams, const int64_t* from, const int64_t* to,
                           const int32_t* d_original_index, int slots, int32_t* h_min_frame, int32_t* h_max_frame);
int cc_engine_scatter_apply(struct cc_engine* e, int stream, int64_t from, int64_t to, const int32_t* d_original_index, int slots,
                            uint8_t* d_is_ground, uint32_t* d_detection, int64_t max_points);

/* ---- host-side pose arithmetic of the same call sites (plain C, no device work) ------------------------------------------ */

/* KittiLoader::interpolate (kitti_loader.cpp:297-328): pose at `stamp` from stamp-sorted poses (slerp + lerp, clamped at the ends). */
int cc_kitti_pose_interpolate(int64_t n_poses, const uint64_t* stamps, const double* poses, uint64_t stamp, double out[12]);
/* The lookup table of undoEgoMotionCorrection (kitti_loader.cpp:183-197). Writes num_bins x 12 doubles to `out` (capacity in
 * bins) and returns num_bins through *num_bins. */
int cc_kitti_bin_transforms(int64_t n_poses, const uint64_t* stamps, const double* poses, uint64_t rotation_start_stamp,
                            uint64_t rotation_end_stamp, const double mid_pose[12], double* out, int32_t capacity, int32_t* num_bins);
/* Stamp (kitti_demo.cpp:132-135) and interpolated odom_from_velodyne (kitti_demo.cpp:388-389) of the 2200 pseudo-firings of a frame. */
int cc_kitti_firing_stamps_and_poses(int64_t n_poses, const uint64_t* stamps, const double* poses, uint64_t start_stamp,
                                     uint64_t end_stamp, uint64_t* out_stamps, double* out_poses);
/* KittiLoader::getStartEndTimestampsVelodyne (kitti_loader.cpp:525-540). */
int cc_kitti_start_end_stamps(int64_t n, const uint64_t* middle, uint64_t* start, uint64_t* end);
/* One line of poses.txt -> odom_from_x (getAllDynamicTransforms, kitti_loader.cpp:330-369): odom_from_first_cam0 * first_cam0_from_cam0 *
 * cam0_from_x, where row12 are the 12 numbers of the line. */
int cc_kitti_pose_from_line(const double row12[12], const double cam0_from_x[12], double out[12]);

#ifdef __cplusplus
}
#endif
#endif
