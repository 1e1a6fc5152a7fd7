// continuous_clustering.hpp — the reference's public C++ API on top of the MI355X C-ABI (include/cc_hip.h).
//
// Mirrors include/continuous_clustering/clustering/continuous_clustering.hpp:24-290 and point_types.hpp:10-28 of
// UniBwTAS/continuous_clustering: same namespace, type names, member names, method names, argument meaning,
// callback signatures and error behaviour (std::runtime_error with the reference's texts), so that front-ends written
// against the reference (src/tools/kitti_demo.cpp:276-313,403, src/ros/continuous_clustering_node.cpp:149-163,
// src/ros/ros_utils.cpp:40-63) compile against this header unchanged. The per-point work happens in the HIP kernels
// behind cc_engine_*; this class is host code only: it batches firings, replays the engine's event log as the
// reference's callbacks (in the single-threaded reference order) and keeps the public `range_image_` mirror current
// for the columns a callback announces.
//
// Poses: the reference takes Eigen::Isometry3d. Eigen is not a dependency here; every pose parameter is a template
// that only needs `tf(row, col)` for row < 3, col < 4 — Eigen::Isometry3d satisfies it, and so does the Pose3d below.
//
// Differences a caller can observe (DESIGN.md "Drop-in boundary"):
//  * with setBatchSize(n > 1) callbacks are delivered when n firings have accumulated (or flush() is called), not
//    inside the addFiring call that caused them; order and content are unchanged. The default n = 1 keeps the
//    reference's synchronous behaviour.
//  * Point::child_points, associated_trees (of root points), number_of_visited_neighbors and the per-tree values are filled from
//    what the engine exports (cc_column_view, tree-link log); visited_at_continuous_azimuth_angle (BFS scratch of cc.cpp:854-894)
//    is not tracked.
//  * is_single_threaded = false (the reference's default, cc.hpp:24-27) is the ASYNCHRONOUS mode: addFiring only enqueues
//    (cc.cpp:92) and a worker thread inside the class hands what has queued up to the engine and runs the callbacks — in the
//    single-threaded order, which is one of the orders the reference's thread pipeline can produce (SURVEY 8b "Threading").
//    reset / setConfiguration / setTransformRobotFrameFromSensorFrame / flush / the destructor wait for the worker to drain.
//    An exception on the worker (the reference: uncaught on a pool thread -> std::terminate) is kept and rethrown by the next
//    call on the caller's thread.
#pragma once

#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#if __has_include(<Eigen/Geometry>)
#include <Eigen/Geometry>
#define CC_AMD_HAVE_EIGEN 1
#endif

#include "../../include/cc_hip.h"

namespace continuous_clustering
{

// ---- general.hpp:84-204 (only what Point needs) -----------------------------------------------------------------
struct Point3D
{
    Point3D() : x(0), y(0), z(0){};
    Point3D(float x, float y, float z) : x(x), y(y), z(z){};
    float x{};
    float y{};
    float z{};
};

// ---- label values: continuous_clustering.hpp:15-22, general.hpp:208-357 ------------------------------------------
enum
{
    GP_UNKNOWN = CC_GP_UNKNOWN,
    GP_GROUND = CC_GP_GROUND,
    GP_OBSTACLE = CC_GP_OBSTACLE,
    GP_EGO_VEHICLE = CC_GP_EGO_VEHICLE,
    GP_FOG = CC_GP_FOG,
    WHITE = CC_DBG_WHITE
};

// ---- point_types.hpp:10-28 ----------------------------------------------------------------------------------------
struct RawPoint
{
    float x{};
    float y{};
    float z{};
    uint64_t firing_index{};
    uint8_t intensity{};
    uint64_t stamp{};
    uint64_t globally_unique_point_index{};
};

struct RawPoints
{
    uint64_t stamp;
    std::vector<RawPoint> points;
    typedef std::shared_ptr<RawPoints> Ptr;
    typedef std::shared_ptr<RawPoints const> ConstPtr;
};

// ---- continuous_clustering.hpp:24-87 ------------------------------------------------------------------------------
struct GeneralConfiguration
{
    bool is_single_threaded{false};
};
struct ContinuousRangeImageConfiguration
{
    bool sensor_is_clockwise{true};
    int num_columns{1700};
    bool supplement_inclination_angle_for_nan_cells{true};
};
struct ContinuousGroundSegmentationConfiguration
{
    float max_slope{0.2};
    float first_ring_as_ground_max_allowed_z_diff{0.4};
    float first_ring_as_ground_min_allowed_z_diff{-0.4};
    float last_ground_point_slope_higher_than{-0.1};
    float last_ground_point_distance_smaller_than{5.};
    float ground_because_close_to_last_certain_ground_max_z_diff{0.4};
    float ground_because_close_to_last_certain_ground_max_dist_diff{2.0};
    float obstacle_because_next_certain_obstacle_max_dist_diff{0.3};
    bool use_terrain{false};
    float terrain_max_allowed_z_diff{0.4};
    float height_ref_to_maximum_{}, height_ref_to_ground_{};
    float length_ref_to_front_end_{}, length_ref_to_rear_end_{};
    float width_ref_to_left_mirror_{}, width_ref_to_right_mirror_{};
    bool fog_filtering_enabled{false};
    uint8_t fog_filtering_intensity_below{2};
    float fog_filtering_distance_below{18};
    float fog_filtering_inclination_above{-0.06};
};
struct ContinuousClusteringConfiguration
{
    float max_distance{0.7};
    int max_steps_in_row{20};
    int max_steps_in_column{20};
    bool stop_after_association_enabled{true};
    int stop_after_association_min_steps{1};
    bool ignore_points_in_chessboard_pattern{true};
    bool ignore_points_with_too_big_inclination_angle_diff{true};
    bool use_last_point_for_cluster_stamp{false};
    int cluster_point_trees_every_nth_column{1};
};
struct Configuration
{
    GeneralConfiguration general{};
    ContinuousRangeImageConfiguration range_image{};
    ContinuousGroundSegmentationConfiguration ground_segmentation{};
    ContinuousClusteringConfiguration clustering{};
};

// ---- continuous_clustering.hpp:89-115 -------------------------------------------------------------------------------
class RangeImageIndex
{
  public:
    RangeImageIndex(uint16_t row_index, int64_t column_index) : column_index(column_index), row_index(row_index)
    {
    }
    bool operator==(const RangeImageIndex& other) const
    {
        return row_index == other.row_index && column_index == other.column_index;
    }
    bool operator!=(const RangeImageIndex& other) const
    {
        return row_index != other.row_index || column_index != other.column_index;
    }
    bool operator<(const RangeImageIndex& other) const
    {
        return row_index < other.row_index || (row_index == other.row_index && column_index < other.column_index);
    }
    int64_t column_index{0};
    uint16_t row_index{0};
};

// ---- continuous_clustering.hpp:126-161 ------------------------------------------------------------------------------
struct Point
{
    Point3D xyz{std::nanf(""), std::nanf(""), std::nanf("")};
    uint64_t firing_index{0};
    uint8_t intensity{0};
    float distance{std::nanf("")};
    float azimuth_angle{std::nanf("")};
    float inclination_angle{std::nanf("")};
    double continuous_azimuth_angle{std::nan("")};
    int64_t global_column_index{-1};
    int local_column_index{-1};
    int row_index{-1};
    uint64_t stamp{0};
    uint64_t globally_unique_point_index{static_cast<uint64_t>(-1)};
    uint8_t ground_point_label{0};
    float height_over_ground{std::nanf("")};
    uint8_t debug_ground_point_label{WHITE};
    bool is_ignored{false};
    double finished_at_continuous_azimuth_angle{0.f};
    std::list<RangeImageIndex> child_points{};
    std::set<RangeImageIndex> associated_trees{};
    RangeImageIndex tree_root_{0, -1};
    uint32_t tree_num_points{0};
    uint32_t cluster_width{0};
    uint64_t tree_id{0};
    uint64_t id{0};
    double visited_at_continuous_azimuth_angle{-1.};
    bool belongs_to_finished_cluster{false};
    int number_of_visited_neighbors{0};
};

// Minimal rigid transform usable wherever the reference takes an Eigen::Isometry3d.
struct Pose3d
{
    double m[12]{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}; // row-major 3x4 [R|t]
    static Pose3d Identity()
    {
        return Pose3d();
    }
    double operator()(int r, int c) const
    {
        return m[r * 4 + c];
    }
    double& operator()(int r, int c)
    {
        return m[r * 4 + c];
    }
};

class ContinuousClustering
{
  public:
    ContinuousClustering();
    ~ContinuousClustering();
    ContinuousClustering(const ContinuousClustering&) = delete;
    ContinuousClustering& operator=(const ContinuousClustering&) = delete;

    // general (continuous_clustering.hpp:205-207)
    void reset(int num_rows);
    void setConfiguration(const Configuration& config);
    bool resetRequired() const;

    // range image generation (continuous_clustering.hpp:210)
    template<class Iso>
    void addFiring(const RawPoints::ConstPtr& firing, const Iso& odom_from_sensor)
    {
        double tf[12];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++)
                tf[r * 4 + c] = odom_from_sensor(r, c);
        addFiringImpl(firing, tf);
    }

    // ground point segmentation (continuous_clustering.hpp:213-214)
    template<class Iso>
    void setTransformRobotFrameFromSensorFrame(const Iso& tf)
    {
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++)
                robot_from_sensor_[r * 4 + c] = tf(r, c);
        setRobotTransformImpl();
    }
    bool hasTransformRobotFrameFromSensorFrame();

#ifdef CC_AMD_HAVE_EIGEN
    // the reference's exact signatures (continuous_clustering.hpp:210,213): preferred over the templates for Eigen::Isometry3d arguments
    void addFiring(const RawPoints::ConstPtr& firing, const Eigen::Isometry3d& odom_from_sensor)
    {
        double tf[12];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++)
                tf[r * 4 + c] = odom_from_sensor(r, c);
        addFiringImpl(firing, tf);
    }
    void setTransformRobotFrameFromSensorFrame(const Eigen::Isometry3d& tf)
    {
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++)
                robot_from_sensor_[r * 4 + c] = tf(r, c);
        setRobotTransformImpl();
    }
#endif

    // continuous clustering (continuous_clustering.hpp:217-218)
    void setFinishedColumnCallback(std::function<void(int64_t, int64_t, bool)> cb);
    void setFinishedClusterCallback(std::function<void(const std::vector<Point>&, uint64_t)> cb);

    // debugging (continuous_clustering.hpp:221)
    void recordJobQueueWorkload(size_t num_jobs_sensor_input);

    // ---- extensions of the MI355X build -----------------------------------------------------------------------------
    void setBatchSize(int firings_per_launch); // default 1: callbacks fire inside addFiring like the reference; 0 = adaptive (below)
    // hand firings to the engine as they queue up behind a running call: at most max_firings (<= 8) per call, none waits longer than
    // max_wait_us for company. Keeps up with a real-time sensor (a call per firing does not: DESIGN.md section 6).
    void setAdaptiveBatching(int max_firings = 8, int max_wait_us = 150);
    void flush();                              // process buffered firings now
    void setDevice(int hip_device);            // before the first reset(); default 0
    // synchronous mode: hand small calls to the engine's resident kernel (no kernel launch per addFiring; it holds one compute unit while firings
    // keep coming and leaves by itself 20 ms after the last one). Default OFF: measured on MI355X boxes of this pool a call is as long either way
    // (40.1 vs 40.3 us p50 — polling a doorbell over PCIe costs what the dispatch costs). Takes effect at the next reset()
    void setResidentKernel(bool on) { use_resident_ = on; }

  public:
    // range image (implemented as ring buffer) — continuous_clustering.hpp:246-251
    int ring_buffer_max_columns{0};
    int num_columns_{};
    int num_rows_{-1};
    std::vector<Point> range_image_{0};
    int64_t ring_buffer_start_global_column_index{};
    int64_t ring_buffer_end_global_column_index{};

  private:
    void addFiringImpl(const RawPoints::ConstPtr& firing, const double tf[12]);
    void bufferFiring(const RawPoints::ConstPtr& firing, const double tf[12]);
    void setRobotTransformImpl();
    void process();
    // asynchronous mode (is_single_threaded = false)
    void workerLoop();
    void startWorker();
    void warmUp();
    void stopWorker();
    void waitIdle();
    void rethrowWorkerError();
    void check(int rc);
    // mirror of range_image_: the columns a call's events name are fetched once (fetchRanges) and applied in callback order, stage by stage
    enum MirrorStage
    {
        STAGE_GROUND = 0, // what the reference's range image holds when the ground-view callback runs (cc.cpp:618-620)
        STAGE_ASSOC = 1,  // ... after associatePointsInColumn of the column (cc.cpp:773-835), which follows that callback directly
        STAGE_FULL = 2    // ... when the cluster-view callback publishes the column (cc.cpp:1087-1089)
    };
    void fetchRanges(std::vector<std::pair<int64_t, int64_t>> want);
    int64_t viewColumn(int64_t g) const;
    void applyColumns(int64_t from, int64_t to, MirrorStage stage);
    void applyCell(int64_t global_column, int row, MirrorStage stage);
    void clearMirrorColumns(int64_t from, int64_t to);
    void collectClusterPoints(const cc_event& e, const int64_t* gcol, const int32_t* row, size_t cnt);
    int64_t globalColumnOfLocal(int64_t local_column) const;
    void toPod(const Configuration& c, cc_config& out) const;

    Configuration config_;
    cc_engine* engine_{nullptr};
    int device_{0};
    int batch_size_{1};
    bool adaptive_{false};
    bool use_resident_{false};
    int max_wait_us_{150};
    std::chrono::steady_clock::time_point first_buffered_at_{}, last_process_end_{};
    bool reset_required_{false};
    bool has_robot_tf_{false};
    double robot_from_sensor_[12]{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    std::function<void(int64_t, int64_t, bool)> finished_column_callback_;
    std::function<void(const std::vector<Point>&, uint64_t)> finished_cluster_callback_;
    // buffered firings (host SoA staging for the C-ABI)
    std::vector<float> buf_xyz_;
    std::vector<uint8_t> buf_int_;
    std::vector<double> buf_pose_;
    int buffered_{0};
    // firings whose points may still sit in the ring: metadata that never leaves the host (stamp, firing_index, ...)
    std::deque<RawPoints::ConstPtr> firing_log_;
    uint64_t firing_log_base_{0}; // sequence number of firing_log_.front()
    uint64_t firings_submitted_{0};
    std::vector<cc_event> events_;
    std::vector<Point> cluster_points_;
    // scratch for cc_engine_read_columns
    std::vector<float> v_x_, v_y_, v_z_, v_d_, v_i_;
    std::vector<double> v_caz_;
    std::vector<int64_t> v_src_, v_rootc_;
    std::vector<int32_t> v_rootr_;
    std::vector<uint8_t> v_g_, v_dbg_, v_ign_;
    std::vector<uint64_t> v_id_;
    std::vector<double> v_fin_;
    std::vector<uint32_t> v_tpts_, v_width_, v_nchild_;
    std::vector<int32_t> v_visits_, v_parr_;
    std::vector<uint8_t> v_finished_;
    std::vector<int64_t> v_parc_;
    struct ViewRange
    {
        int64_t from, to;
        size_t offset; // first column of the range in the v_* arrays
    };
    std::vector<ViewRange> v_ranges_; // global columns held by the v_* arrays
    std::vector<uint8_t> mirror_stage_; // per ring cell: 0 cleared, 1 ground stage applied, 2 association stage, 3 full
    // Point::associated_trees of the unfinished trees, rebuilt from the engine's tree-link log: root (row, local column) -> linked roots
    std::map<RangeImageIndex, std::set<RangeImageIndex>> tree_links_;
    std::vector<int64_t> link_buf_;
    std::vector<int64_t> col_min_src_; // per ring column: oldest firing that still has a point in it (-1 unknown)
    std::list<size_t> num_pending_jobs_;
    // asynchronous mode: firings queue here (cc.cpp:92) and the worker thread takes what has queued up
    struct QueuedFiring
    {
        RawPoints::ConstPtr firing;
        std::array<double, 12> tf;
    };
    bool async_{false};
    std::thread worker_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_idle_;
    std::deque<QueuedFiring> queue_;
    bool stop_{false};
    bool busy_{false};
    struct TraceEntry
    {
        double at_ms;
        int n;
        double us;
    };
    bool trace_{false}; // CC_ASYNC_TRACE: the worker logs its hand-overs
    std::chrono::steady_clock::time_point trace_t0_;
    std::vector<TraceEntry> trace_log_;
    std::exception_ptr worker_error_;
    std::atomic<bool> reset_required_async_{false};
};

} // namespace continuous_clustering
