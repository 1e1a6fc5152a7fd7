// continuous_clustering.cpp — host-side implementation of the reference's class API over the C-ABI (see the header).
#include "continuous_clustering.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace continuous_clustering
{

ContinuousClustering::ContinuousClustering() = default;

ContinuousClustering::~ContinuousClustering()
{
    try
    {
        stopWorker(); // drains the queue first (callbacks still run), then joins
    }
    catch (...)
    {
    }
    if (trace_ && !trace_log_.empty())
    {
        std::vector<TraceEntry> v = trace_log_;
        std::sort(v.begin(), v.end(), [](const TraceEntry& a, const TraceEntry& b) { return a.us > b.us; });
        fprintf(stderr, "[cc async trace] %zu hand-overs; the longest:", v.size());
        for (size_t i = 0; i < v.size() && i < 8; i++)
            fprintf(stderr, " (t %.1f ms, %d firings, %.0f us)", v[i].at_ms, v[i].n, v[i].us);
        fprintf(stderr, "\n");
    }
    if (engine_)
        cc_engine_destroy(engine_);
}

// ---- asynchronous mode: is_single_threaded = false (cc.hpp:24-27, cc.cpp:49-63,92; thread_pool.hpp:58-67) -------------------------
// The reference enqueues the firing and returns; five pools of worker threads run the stages. Here ONE worker thread takes whatever
// has queued up behind the engine call in flight (a handful of firings at sensor rate: one captured-graph launch; everything that is
// there when the caller runs ahead) and replays the events as callbacks, in the single-threaded order.
void ContinuousClustering::startWorker()
{
    if (worker_.joinable())
        return;
    stop_ = false;
    busy_ = true; // (until the worker's warm-up call on the engine has returned: workerLoop)
    worker_ = std::thread([this] { workerLoop(); });
}

void ContinuousClustering::stopWorker()
{
    if (!worker_.joinable())
        return;
    // a callback runs ON the worker: reset() / setConfiguration() / flush() from inside one would wait for the worker to become idle, or join the
    // calling thread itself — a deadlock. The reference's callbacks run on pool threads that reset() shuts down too; say so instead of hanging.
    if (std::this_thread::get_id() == worker_.get_id())
        throw std::runtime_error("ContinuousClustering: reset / setConfiguration / setTransformRobotFrameFromSensorFrame / flush called from inside a callback "
                                 "of the asynchronous mode (the callback runs on the worker thread it would have to wait for)");
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_idle_.wait(lk, [this] { return (queue_.empty() && !busy_) || worker_error_; });
        stop_ = true;
    }
    cv_work_.notify_all();
    worker_.join();
}

void ContinuousClustering::waitIdle()
{
    if (!worker_.joinable())
        return;
    if (std::this_thread::get_id() == worker_.get_id())
        throw std::runtime_error("ContinuousClustering: reset / setConfiguration / setTransformRobotFrameFromSensorFrame / flush called from inside a callback "
                                 "of the asynchronous mode (the callback runs on the worker thread it would have to wait for)");
    std::unique_lock<std::mutex> lk(mu_);
    cv_idle_.wait(lk, [this] { return (queue_.empty() && !busy_) || worker_error_; });
}

void ContinuousClustering::rethrowWorkerError()
{
    std::exception_ptr e;
    {
        std::lock_guard<std::mutex> lk(mu_);
        e = worker_error_;
        worker_error_ = nullptr;
        if (e)
            queue_.clear(); // (the reference would have terminated; what queued up behind the failure is dropped)
    }
    if (e)
        std::rethrow_exception(e);
}

void ContinuousClustering::workerLoop()
{
    std::vector<QueuedFiring> take;
    trace_ = getenv("CC_ASYNC_TRACE") != nullptr;
    trace_t0_ = std::chrono::steady_clock::now();
    // the first HIP call of a host thread sets the thread up with the runtime (milliseconds): here, not in front of the first firing.
    // busy_ is set by startWorker() before the thread exists and cleared here: a caller that goes through waitIdle() right behind reset()
    // (setTransformRobotFrameFromSensorFrame) waits for this call instead of racing with it on the engine, which is not thread-safe.
    if (engine_)
        (void) cc_engine_sync(engine_);
    {
        std::lock_guard<std::mutex> lk(mu_);
        busy_ = false;
    }
    cv_idle_.notify_all();
    while (true)
    {
        take.clear();
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_work_.wait(lk, [this] { return stop_ || (!queue_.empty() && !worker_error_); });
            if (stop_)
                return;
            // at sensor rate a few firings queue up per engine call: at most 8 go into one captured-graph launch (~0.1 ms); a caller that
            // runs ahead (replay from disk) gets large batches, bounded by what a call may publish (include/cc_hip.h)
            size_t n = queue_.size();
            const size_t big = static_cast<size_t>(std::max(64, num_columns_));
            n = n <= 8 ? n : std::min(n, big);
            for (size_t i = 0; i < n; i++)
            {
                take.push_back(std::move(queue_.front()));
                queue_.pop_front();
            }
            busy_ = true;
        }
        const auto tr0 = std::chrono::steady_clock::now();
        try
        {
            for (const QueuedFiring& q : take)
                bufferFiring(q.firing, q.tf.data());
            process();
        }
        catch (...)
        {
            std::lock_guard<std::mutex> lk(mu_);
            worker_error_ = std::current_exception();
            buffered_ = 0;
            // (the failed batch's firings are still in firing_log_, which the mirror indexes by firing number: after the error has been rethrown to
            // the caller the only consistent way on is reset(), as after the reference's own soft error, cc.cpp:252-261)
            reset_required_async_ = true;
        }
        if (trace_)
        {
            // CC_ASYNC_TRACE=1: the longest hand-overs of the run (how many firings, how long the engine call and the callbacks took, when)
            const auto tr1 = std::chrono::steady_clock::now();
            const double us = std::chrono::duration<double, std::micro>(tr1 - tr0).count();
            trace_log_.push_back({std::chrono::duration<double, std::milli>(tr0 - trace_t0_).count(), (int) take.size(), us});
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            busy_ = false;
        }
        cv_idle_.notify_all();
    }
}

void ContinuousClustering::toPod(const Configuration& c, cc_config& o) const
{
    cc_config_default(&o);
    o.is_single_threaded = c.general.is_single_threaded;
    o.sensor_is_clockwise = c.range_image.sensor_is_clockwise;
    o.num_columns = c.range_image.num_columns;
    o.supplement_inclination_angle_for_nan_cells = c.range_image.supplement_inclination_angle_for_nan_cells;
    const auto& g = c.ground_segmentation;
    o.max_slope = g.max_slope;
    o.first_ring_as_ground_max_allowed_z_diff = g.first_ring_as_ground_max_allowed_z_diff;
    o.first_ring_as_ground_min_allowed_z_diff = g.first_ring_as_ground_min_allowed_z_diff;
    o.last_ground_point_slope_higher_than = g.last_ground_point_slope_higher_than;
    o.last_ground_point_distance_smaller_than = g.last_ground_point_distance_smaller_than;
    o.ground_because_close_to_last_certain_ground_max_z_diff = g.ground_because_close_to_last_certain_ground_max_z_diff;
    o.ground_because_close_to_last_certain_ground_max_dist_diff = g.ground_because_close_to_last_certain_ground_max_dist_diff;
    o.obstacle_because_next_certain_obstacle_max_dist_diff = g.obstacle_because_next_certain_obstacle_max_dist_diff;
    o.use_terrain = g.use_terrain;
    o.terrain_max_allowed_z_diff = g.terrain_max_allowed_z_diff;
    o.height_ref_to_maximum_ = g.height_ref_to_maximum_;
    o.height_ref_to_ground_ = g.height_ref_to_ground_;
    o.length_ref_to_front_end_ = g.length_ref_to_front_end_;
    o.length_ref_to_rear_end_ = g.length_ref_to_rear_end_;
    o.width_ref_to_left_mirror_ = g.width_ref_to_left_mirror_;
    o.width_ref_to_right_mirror_ = g.width_ref_to_right_mirror_;
    o.fog_filtering_enabled = g.fog_filtering_enabled;
    o.fog_filtering_intensity_below = g.fog_filtering_intensity_below;
    o.fog_filtering_distance_below = g.fog_filtering_distance_below;
    o.fog_filtering_inclination_above = g.fog_filtering_inclination_above;
    const auto& k = c.clustering;
    o.max_distance = k.max_distance;
    o.max_steps_in_row = k.max_steps_in_row;
    o.max_steps_in_column = k.max_steps_in_column;
    o.stop_after_association_enabled = k.stop_after_association_enabled;
    o.stop_after_association_min_steps = k.stop_after_association_min_steps;
    o.ignore_points_in_chessboard_pattern = k.ignore_points_in_chessboard_pattern;
    o.ignore_points_with_too_big_inclination_angle_diff = k.ignore_points_with_too_big_inclination_angle_diff;
    o.use_last_point_for_cluster_stamp = k.use_last_point_for_cluster_stamp;
    o.cluster_point_trees_every_nth_column = k.cluster_point_trees_every_nth_column;
}

// Status codes -> the reference's exceptions (continuous_clustering.cpp:90-91, :298-299, :337-344, :1052, :1073-1075).
void ContinuousClustering::check(int rc)
{
    if (rc == CC_OK)
        return;
    cc_stream_state st{};
    if (engine_)
        cc_engine_stream_state(engine_, 0, &st);
    switch (rc)
    {
        case CC_ERR_NO_ROBOT_TRANSFORM:
            throw std::runtime_error("Transform robot frame from sensor frame was not set yet!");
        case CC_ERR_RING_OVERRUN:
            throw std::runtime_error(
                "This column is not cleared. Probably this means the ring buffer is full or there "
                "is some other issue with clearing (not cleared at all or written after clearing): " +
                std::to_string(st.error_a) + ", " + std::to_string(st.error_b) + ", " + std::to_string(ring_buffer_max_columns) +
                "; This typically happens when the clustering is not fast enough to handle all the firings. Consider "
                "to play the sensor data more slowly or to adjust the parameters to make the clustering faster.");
        case CC_ERR_BOOKKEEPING:
            throw std::runtime_error("This shouldn't happen, ring buffer is not allowed to increase at the front: " +
                                     std::to_string(st.error_a) + ", " + std::to_string(st.error_b));
        case CC_ERR_NO_DEVICE:
            throw std::runtime_error("continuous_clustering_amd: no MI355X (gfx950) device is visible; there is no CPU path");
        default:
            throw std::runtime_error(std::string("continuous_clustering_amd: engine error ") + std::to_string(rc) + ": " +
                                     (engine_ ? cc_engine_last_error(engine_) : ""));
    }
}

// a wall 10 m around the sensor, fed in calls of every size class (1 .. 8 firings: one captured graph each; < 64: the small-call kernels; a few
// hundred: the block-parallel insertion): results are thrown away
void ContinuousClustering::warmUp()
{
    const int R = num_rows_, C = num_columns_;
    const int sizes[] = {1, 2, 3, 4, 5, 6, 7, 8, 40, 200, 1, 300};
    int total = 0;
    for (int n : sizes)
        total += n;
    std::vector<float> xyz(static_cast<size_t>(total) * R * 3);
    std::vector<uint8_t> inten(static_cast<size_t>(total) * R, 100);
    std::vector<double> poses(static_cast<size_t>(total) * 12, 0.);
    const double width = 2 * M_PI / C;
    for (int f = 0; f < total; f++)
    {
        const double a = (f + 0.5) * width;
        const double az = config_.range_image.sensor_is_clockwise ? M_PI - a : -M_PI + a;
        for (int r = 0; r < R; r++)
        {
            const double incl = (2.0 - 26.0 * r / std::max(1, R - 1)) * M_PI / 180.0;
            float* p = &xyz[(static_cast<size_t>(f) * R + r) * 3];
            p[0] = static_cast<float>(10.0 * std::cos(incl) * std::cos(az));
            p[1] = static_cast<float>(10.0 * std::cos(incl) * std::sin(az));
            p[2] = static_cast<float>(10.0 * std::sin(incl));
        }
        double* T = &poses[static_cast<size_t>(f) * 12];
        T[0] = T[5] = T[10] = 1.0;
    }
    const double id[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    (void) cc_engine_set_robot_from_sensor(engine_, 0, id);
    int f = 0;
    for (int n : sizes)
    {
        if (cc_engine_add_firings(engine_, 0, n, &xyz[static_cast<size_t>(f) * R * 3], &inten[static_cast<size_t>(f) * R], &poses[static_cast<size_t>(f) * 12]) != CC_OK)
            break;
        f += n;
    }
    (void) cc_engine_sync(engine_);
}

// ---- continuous_clustering.cpp:11-64 ----------------------------------------------------------------------------------
void ContinuousClustering::reset(int num_rows)
{
    stopWorker(); // cc.cpp:20-31: the reference shuts its pools down (after their queues have drained) and starts new ones below
    {
        std::lock_guard<std::mutex> lk(mu_);
        queue_.clear();
        worker_error_ = nullptr;
    }
    num_columns_ = config_.range_image.num_columns;
    num_rows_ = num_rows;
    ring_buffer_max_columns = num_columns_ * 10;
    cc_config pod;
    toPod(config_, pod);
    const bool created = !engine_;
    if (!engine_)
        check(cc_engine_create(&engine_, device_, 1, num_rows, &pod));
    else
    {
        check(cc_engine_set_config(engine_, &pod));
        check(cc_engine_reset(engine_, num_rows));
    }
    check(cc_engine_record_events(engine_, 1));
    if (!config_.general.is_single_threaded && created)
    {
        // Asynchronous mode = a live sensor behind the front-end: everything a call pays only the first time (the first launch of every captured
        // graph and of every kernel of the larger call sizes: 8 - 9 ms each, twice, inside the first 25 ms of a stream at 22 000 firings/s) is
        // paid here, once per object (where its first reset() finds the asynchronous mode configured), on a made-up quarter rotation; the engine is reset behind it.
        warmUp(); // (only on the engine this call created: it has seen nothing else)
        check(cc_engine_reset(engine_, num_rows));
        check(cc_engine_set_option(engine_, "forget_inclination_table", 1)); // (what reset keeps across calls, cc.cpp:46, must not keep the made-up data's)
        check(cc_engine_record_events(engine_, 1));
    }
    // Synchronous mode (is_single_threaded, thread_pool.hpp:58-64: addFiring returns when the firing's stages have run — kitti_demo.cpp:280): on request
    // (setResidentKernel) calls of a few firings are handed to the engine's RESIDENT kernel through a doorbell in pinned memory instead of a kernel
    // launch each; the column reads of the mirror run beside it. (The asynchronous mode's worker takes whatever has queued up: graphs of 1 .. 8 firings
    // and larger calls.)
    (void) cc_engine_set_option(engine_, "resident", (config_.general.is_single_threaded && use_resident_) ? 1 : 0);
    // the graphs of calls of 1 .. 8 firings: built here, not in front of live data — and LAST: cc_engine_reset, cc_engine_record_events and every
    // cc_engine_set_option drop captured graphs (they bake configuration, geometry and plane pointers)
    (void) cc_engine_set_option(engine_, "prewarm_small_graphs", 1);
    range_image_.assign(static_cast<size_t>(ring_buffer_max_columns) * num_rows, Point{});
    for (auto& p : range_image_)
        p.ground_point_label = GP_UNKNOWN; // clearColumns (cc.cpp:1125)
    ring_buffer_start_global_column_index = -1;
    ring_buffer_end_global_column_index = -1;
    reset_required_ = false;
    has_robot_tf_ = false; // cc.cpp:39
    buffered_ = 0;
    firing_log_.clear();
    firing_log_base_ = 0;
    firings_submitted_ = 0;
    buf_xyz_.clear();
    buf_int_.clear();
    buf_pose_.clear();
    col_min_src_.assign(static_cast<size_t>(ring_buffer_max_columns), -1);
    mirror_stage_.assign(range_image_.size(), 0);
    tree_links_.clear();
    v_ranges_.clear();
    reset_required_async_ = false;
    async_ = !config_.general.is_single_threaded; // cc.cpp:49-63
    if (async_)
        startWorker();
}

// ---- continuous_clustering.cpp:66-81 ----------------------------------------------------------------------------------
void ContinuousClustering::setConfiguration(const Configuration& config)
{
    if (config_.general.is_single_threaded != config.general.is_single_threaded)
        reset_required_ = true;
    if (config_.range_image.sensor_is_clockwise != config.range_image.sensor_is_clockwise)
        reset_required_ = true;
    if (config_.range_image.num_columns != config.range_image.num_columns)
        reset_required_ = true;
    if (engine_)
        flush(); // (asynchronous mode: the worker drains first — it reads config_)
    config_ = config;
    if (engine_ && num_rows_ > 0 && config.range_image.num_columns == num_columns_)
    {
        cc_config pod;
        toPod(config_, pod);
        check(cc_engine_set_config(engine_, &pod));
    }
}

bool ContinuousClustering::resetRequired() const
{
    return reset_required_ || reset_required_async_.load();
}

void ContinuousClustering::setRobotTransformImpl()
{
    has_robot_tf_ = true;
    if (engine_)
    {
        flush();
        check(cc_engine_set_robot_from_sensor(engine_, 0, robot_from_sensor_));
    }
}

bool ContinuousClustering::hasTransformRobotFrameFromSensorFrame()
{
    return has_robot_tf_;
}

void ContinuousClustering::setFinishedColumnCallback(std::function<void(int64_t, int64_t, bool)> cb)
{
    finished_column_callback_ = std::move(cb);
}

void ContinuousClustering::setFinishedClusterCallback(std::function<void(const std::vector<Point>&, uint64_t)> cb)
{
    finished_cluster_callback_ = std::move(cb);
}

void ContinuousClustering::recordJobQueueWorkload(size_t num_jobs_sensor_input)
{
    // the reference records 6 queue depths per call (cc.cpp:1147-1159); the stage queues do not exist here
    num_pending_jobs_.push_back(num_jobs_sensor_input);
    for (int i = 0; i < 5; i++)
        num_pending_jobs_.push_back(static_cast<size_t>(buffered_));
    while (num_pending_jobs_.size() > 100000 * 6)
        num_pending_jobs_.pop_front();
}

void ContinuousClustering::setBatchSize(int firings_per_launch)
{
    flush();
    adaptive_ = firings_per_launch <= 0;
    batch_size_ = adaptive_ ? 8 : firings_per_launch; // 8 = what one captured-graph launch of the engine takes (cc_engine_add_firings)
}

// Adaptive batching (setBatchSize(0)): a firing is handed to the engine together with the firings that arrived while the previous
// call was running — at most 8 per call (one graph launch, ~0.1 ms) and never later than `max_wait_us` after the oldest buffered
// firing arrived. A sensor that delivers 22 000 firings per second is then followed in real time (a call per firing takes longer
// than the firing period); the callbacks' order is the reference's, they are only delivered a few firings later.
void ContinuousClustering::setAdaptiveBatching(int max_firings, int max_wait_us)
{
    flush();
    adaptive_ = true;
    batch_size_ = std::max(1, std::min(8, max_firings));
    max_wait_us_ = std::max(0, max_wait_us);
}

void ContinuousClustering::setDevice(int hip_device)
{
    device_ = hip_device;
}

// ---- continuous_clustering.cpp:88-93 ----------------------------------------------------------------------------------
void ContinuousClustering::addFiringImpl(const RawPoints::ConstPtr& firing, const double tf[12])
{
    if (static_cast<size_t>(num_rows_) != firing->points.size())
        throw std::runtime_error("The number of points in a firing has changed. This is probably a bug!");
    if (!engine_)
        throw std::runtime_error("continuous_clustering_amd: reset(num_rows) must be called before addFiring");
    if (async_)
    {
        // cc.cpp:92: enqueue and return; the stages run on the worker
        rethrowWorkerError();
        {
            std::lock_guard<std::mutex> lk(mu_);
            QueuedFiring q;
            q.firing = firing;
            std::memcpy(q.tf.data(), tf, sizeof(double) * 12);
            queue_.push_back(std::move(q));
        }
        cv_work_.notify_one();
        return;
    }
    bufferFiring(firing, tf);
    // never pass more than 2 * num_columns firings per engine call: everything a call publishes stays readable until the
    // next call (include/cc_hip.h, cc_engine_add_firings)
    if (adaptive_)
    {
        const auto now = std::chrono::steady_clock::now();
        if (buffered_ == 1)
            first_buffered_at_ = now;
        // flush when the call is full, when the oldest firing has waited long enough, or when the caller is not ahead of us (the previous
        // call returned more than a firing period ago: nothing is queued behind this firing)
        const double waited_us = std::chrono::duration<double, std::micro>(now - first_buffered_at_).count();
        const double idle_us = std::chrono::duration<double, std::micro>(now - last_process_end_).count();
        if (buffered_ >= batch_size_ || waited_us >= max_wait_us_ || idle_us >= max_wait_us_)
        {
            process();
            last_process_end_ = std::chrono::steady_clock::now();
        }
        return;
    }
    if (buffered_ >= batch_size_ || buffered_ >= 2 * num_columns_)
        process();
}

void ContinuousClustering::bufferFiring(const RawPoints::ConstPtr& firing, const double tf[12])
{
    const size_t R = static_cast<size_t>(num_rows_);
    buf_xyz_.resize((buffered_ + 1) * R * 3);
    buf_int_.resize((buffered_ + 1) * R);
    buf_pose_.resize((buffered_ + 1) * 12);
    float* x = buf_xyz_.data() + buffered_ * R * 3;
    uint8_t* in = buf_int_.data() + buffered_ * R;
    for (size_t r = 0; r < R; r++)
    {
        const RawPoint& p = firing->points[r];
        x[r * 3 + 0] = p.x;
        x[r * 3 + 1] = p.y;
        x[r * 3 + 2] = p.z;
        in[r] = p.intensity;
    }
    std::memcpy(buf_pose_.data() + buffered_ * 12, tf, 12 * sizeof(double));
    firing_log_.push_back(firing);
    buffered_++;
}

void ContinuousClustering::flush()
{
    if (async_)
    {
        waitIdle();
        rethrowWorkerError();
        return;
    }
    if (buffered_ > 0)
        process();
}

void ContinuousClustering::clearMirrorColumns(int64_t from, int64_t to)
{
    for (int64_t g = std::max<int64_t>(from, 0); g <= to; g++)
    {
        const size_t lc = static_cast<size_t>(g % ring_buffer_max_columns);
        for (int r = 0; r < num_rows_; r++)
        {
            Point& p = range_image_[lc * num_rows_ + r];
            p = Point{};
            p.ground_point_label = GP_UNKNOWN;
            mirror_stage_[lc * num_rows_ + r] = 0;
        }
    }
}

int64_t ContinuousClustering::globalColumnOfLocal(int64_t local_column) const
{
    // the latest global column <= the ring's end that maps to this ring slot
    const int64_t end = ring_buffer_end_global_column_index;
    const int64_t rc = ring_buffer_max_columns;
    return end - (((end % rc) - local_column + rc) % rc);
}

// The columns of the given ranges into the v_* arrays (every field of cc_column_view): ranges merged where they touch, one engine read per merged
// range (cut at 8 columns: what the engine serves from the views a small call mirrored with its results).
void ContinuousClustering::fetchRanges(std::vector<std::pair<int64_t, int64_t>> want)
{
    v_ranges_.clear();
    std::sort(want.begin(), want.end());
    std::vector<std::pair<int64_t, int64_t>> merged;
    for (const auto& r : want)
    {
        if (r.second < r.first)
            continue;
        if (!merged.empty() && r.first <= merged.back().second + 1)
            merged.back().second = std::max(merged.back().second, r.second);
        else
            merged.push_back(r);
    }
    if (!merged.empty())
    {
        // (never more than the ring holds: the oldest columns of an over-long range are gone)
        const int64_t hi = merged.back().second;
        for (auto& r : merged)
            r.first = std::max(r.first, hi - ring_buffer_max_columns + 1);
    }
    size_t total = 0;
    for (const auto& r : merged)
        if (r.second >= r.first)
        {
            v_ranges_.push_back({r.first, r.second, total});
            total += static_cast<size_t>(r.second - r.first + 1);
        }
    const size_t R = static_cast<size_t>(num_rows_);
    const size_t n = total * R;
    v_x_.resize(n), v_y_.resize(n), v_z_.resize(n), v_d_.resize(n), v_i_.resize(n), v_caz_.resize(n), v_src_.resize(n);
    v_rootc_.resize(n), v_rootr_.resize(n), v_g_.resize(n), v_dbg_.resize(n), v_ign_.resize(n), v_id_.resize(n);
    v_fin_.resize(n), v_tpts_.resize(n), v_width_.resize(n), v_nchild_.resize(n), v_visits_.resize(n), v_parr_.resize(n);
    v_finished_.resize(n), v_parc_.resize(n);
    if (v_ranges_.empty())
        return;
    auto view_at = [&](size_t col_offset)
    {
        const size_t o = col_offset * R;
        cc_column_view v{};
        v.x = v_x_.data() + o, v.y = v_y_.data() + o, v.z = v_z_.data() + o, v.distance = v_d_.data() + o;
        v.inclination_angle = v_i_.data() + o, v.continuous_azimuth_angle = v_caz_.data() + o, v.source_firing = v_src_.data() + o;
        v.ground_point_label = v_g_.data() + o, v.debug_ground_point_label = v_dbg_.data() + o, v.is_ignored = v_ign_.data() + o;
        v.id = v_id_.data() + o, v.tree_root_global_column = v_rootc_.data() + o, v.tree_root_row = v_rootr_.data() + o;
        v.finished_at_continuous_azimuth_angle = v_fin_.data() + o, v.tree_num_points = v_tpts_.data() + o;
        v.cluster_width = v_width_.data() + o, v.number_of_visited_neighbors = v_visits_.data() + o;
        v.belongs_to_finished_cluster = v_finished_.data() + o, v.tree_parent_global_column = v_parc_.data() + o;
        v.tree_parent_row = v_parr_.data() + o;
        return v;
    };
    // ONE engine call for all ranges (cc_engine_read_column_ranges: the mirrored views of a small call, else one launch and one copy), eight ranges
    // at a time
    for (size_t k = 0; k < v_ranges_.size(); k += 8)
    {
        const int nr = static_cast<int>(std::min<size_t>(8, v_ranges_.size() - k));
        int64_t fr[8], to[8];
        for (int i = 0; i < nr; i++)
            fr[i] = v_ranges_[k + i].from, to[i] = v_ranges_[k + i].to;
        const cc_column_view v = view_at(v_ranges_[k].offset);
        check(cc_engine_read_column_ranges(engine_, 0, nr, fr, to, &v));
    }
}

// index of global column g in the v_* arrays (in columns), -1: not fetched by the last call of fetchRanges
int64_t ContinuousClustering::viewColumn(int64_t g) const
{
    for (const ViewRange& vr : v_ranges_)
        if (g >= vr.from && g <= vr.to)
            return static_cast<int64_t>(vr.offset) + (g - vr.from);
    return -1;
}

// Bring the mirror cells of the columns [from, to] (inside the fetched range) to `stage`. STAGE_GROUND: range-image and ground
// segmentation fields, clustering fields as clearColumns left them (the ground-view callback runs before the column is associated,
// cc.cpp:618-623). STAGE_ASSOC: tree root, visited-neighbour count, and the point enters its parent's child list (cc.cpp:663) —
// columns reach this stage in ascending order and rows ascend inside a column, which is the order the reference attaches points in.
// STAGE_FULL: cluster id and the per-tree values of root points.
void ContinuousClustering::applyColumns(int64_t from, int64_t to, MirrorStage stage)
{
    const size_t R = static_cast<size_t>(num_rows_);
    for (int64_t g = from; g <= to; g++)
    {
        const int64_t vc = viewColumn(g);
        if (vc < 0)
            continue;
        const size_t lc = static_cast<size_t>(g % ring_buffer_max_columns);
        int64_t min_src = -1;
        for (size_t r = 0; r < R; r++)
        {
            const size_t i = static_cast<size_t>(vc) * R + r;
            if (v_src_[i] >= 0 && (min_src < 0 || v_src_[i] < min_src))
                min_src = v_src_[i];
            applyCell(g, static_cast<int>(r), stage);
        }
        col_min_src_[lc] = min_src;
    }
}

void ContinuousClustering::applyCell(int64_t g, int row, MirrorStage stage)
{
    const int64_t vc = viewColumn(g);
    if (vc < 0)
        return;
    const size_t R = static_cast<size_t>(num_rows_), r = static_cast<size_t>(row);
    const size_t lc = static_cast<size_t>(g % ring_buffer_max_columns);
    const size_t i = static_cast<size_t>(vc) * R + r;
    const size_t ci = lc * R + r;
    Point& p = range_image_[ci];
    if (mirror_stage_[ci] < 1)
    {
        p.xyz = Point3D(v_x_[i], v_y_[i], v_z_[i]);
        p.distance = v_d_[i];
        p.inclination_angle = v_i_[i];
        p.continuous_azimuth_angle = v_caz_[i];
        p.global_column_index = g; // refilled for every cell of a segmented column (cc.cpp:348-350)
        p.local_column_index = static_cast<int>(lc);
        p.ground_point_label = v_g_[i];
        p.debug_ground_point_label = v_dbg_[i];
        p.is_ignored = v_ign_[i] != 0;
        const int64_t src = v_src_[i];
        if (src >= static_cast<int64_t>(firing_log_base_) && src < static_cast<int64_t>(firing_log_base_ + firing_log_.size()))
        {
            const RawPoint& raw = firing_log_[static_cast<size_t>(src - firing_log_base_)]->points[r];
            p.row_index = static_cast<int>(r);
            p.firing_index = raw.firing_index;
            p.intensity = raw.intensity;
            p.stamp = raw.stamp;
            p.globally_unique_point_index = raw.globally_unique_point_index;
            p.azimuth_angle = std::atan2(raw.y, raw.x); // cc.cpp:142 (sensor frame)
        }
        else
        {
            p.row_index = -1;
            p.firing_index = 0;
            p.intensity = 0;
            p.stamp = 0;
            p.globally_unique_point_index = static_cast<uint64_t>(-1);
            p.azimuth_angle = std::nanf("");
        }
        mirror_stage_[ci] = 1;
    }
    if (stage >= STAGE_ASSOC && mirror_stage_[ci] < 2)
    {
        // what associatePointsInColumn leaves in the point (cc.cpp:661-663, 725, 814-815): final once the column is associated
        if (v_rootc_[i] >= 0)
        {
            p.tree_root_ = RangeImageIndex(static_cast<uint16_t>(v_rootr_[i]), v_rootc_[i] % ring_buffer_max_columns);
            p.tree_id = static_cast<uint64_t>(v_rootc_[i]) * R + static_cast<uint64_t>(v_rootr_[i]); // cc.cpp:662,815
        }
        else
        {
            p.tree_root_ = RangeImageIndex(0, -1);
            p.tree_id = 0;
        }
        p.number_of_visited_neighbors = v_visits_[i];
        if (v_parc_[i] >= 0)
        {
            const size_t plc = static_cast<size_t>(v_parc_[i] % ring_buffer_max_columns);
            Point& q = range_image_[plc * R + static_cast<size_t>(v_parr_[i])];
            if (q.global_column_index == v_parc_[i]) // (the parent's column is still in the mirror)
                q.child_points.emplace_back(static_cast<uint16_t>(r), static_cast<int64_t>(lc));
        }
        mirror_stage_[ci] = 2;
    }
    if (stage == STAGE_FULL && mirror_stage_[ci] < 3)
    {
        // values that keep changing until the point's tree is finished: the cluster id (cc.cpp:1005) and, in the root point, the
        // per-tree values (cc.cpp:666-671, 818-822) and belongs_to_finished_cluster (:933). Final for every published column.
        p.id = v_id_[i];
        p.finished_at_continuous_azimuth_angle = v_fin_[i];
        p.tree_num_points = v_tpts_[i];
        p.cluster_width = v_width_[i];
        p.belongs_to_finished_cluster = v_finished_[i] != 0;
        mirror_stage_[ci] = 3;
    }
}

// cluster_points of collectPointsForCusterAndPublish (cc.cpp:985-1016) for one finished cluster: its member cells (gathered on the
// device in column-then-row order) are brought to the full stage, then the trees are walked in the order of cc.cpp:851-910 (breadth
// first over Point::associated_trees from the cluster's oldest tree) and the points of each tree breadth first over the child lists.
void ContinuousClustering::collectClusterPoints(const cc_event& e, const int64_t* gcol, const int32_t* row, size_t cnt)
{
    cluster_points_.clear();
    const size_t R = static_cast<size_t>(num_rows_);
    const int64_t rc = ring_buffer_max_columns;
    std::vector<RangeImageIndex> roots; // in creation order (column, then row)
    for (size_t k = 0; k < cnt; k++)
    {
        applyCell(gcol[k], row[k], STAGE_FULL); // members only, in column-then-row order: the order their parents' child lists grow in
        const Point& p = range_image_[static_cast<size_t>(gcol[k] % rc) * R + static_cast<size_t>(row[k])];
        if (p.tree_root_.column_index == gcol[k] % rc && p.tree_root_.row_index == row[k])
            roots.emplace_back(static_cast<uint16_t>(row[k]), gcol[k] % rc);
    }
    std::list<RangeImageIndex> trees_to_visit, trees;
    std::set<RangeImageIndex> visited;
    if (!roots.empty())
        trees_to_visit.push_back(roots.front());
    while (!trees_to_visit.empty())
    {
        const RangeImageIndex t = trees_to_visit.front();
        trees_to_visit.pop_front();
        if (!visited.insert(t).second)
            continue;
        trees.push_back(t);
        auto it = tree_links_.find(t);
        if (it != tree_links_.end())
            for (const RangeImageIndex& o : it->second)
                if (!visited.count(o))
                    trees_to_visit.push_back(o);
    }
    for (const RangeImageIndex& r : roots) // (a root the link walk did not reach would mean a lost log entry: keep its points)
        if (!visited.count(r))
            trees.push_back(r);
    std::list<RangeImageIndex> points_to_visit;
    for (const RangeImageIndex& t : trees)
    {
        points_to_visit.clear();
        points_to_visit.push_back(t);
        while (!points_to_visit.empty())
        {
            const RangeImageIndex cur = points_to_visit.front();
            points_to_visit.pop_front();
            Point& p = range_image_[static_cast<size_t>(cur.column_index) * R + cur.row_index];
            p.id = e.c; // cc.cpp:1005
            cluster_points_.push_back(p);
            for (const RangeImageIndex& ch : p.child_points)
                points_to_visit.push_back(ch);
        }
        tree_links_.erase(t);
    }
}

void ContinuousClustering::process()
{
    const int n = buffered_;
    buffered_ = 0;
    const int rc = cc_engine_add_firings(engine_, 0, n, buf_xyz_.data(), buf_int_.data(), buf_pose_.data());
    firings_submitted_ += static_cast<uint64_t>(n);
    cc_stream_state st{};
    if (cc_engine_stream_state(engine_, 0, &st) == CC_OK)
    {
        if (st.reset_required)
            reset_required_async_ = true; // cc.cpp:252-261 (set on the worker thread in the asynchronous mode)
        ring_buffer_end_global_column_index = st.ring_buffer_end_global_column_index;
    }
    check(rc);

    int64_t pending = 0;
    check(cc_engine_pending_events(engine_, 0, &pending));
    events_.resize(static_cast<size_t>(pending));
    int64_t got = 0;
    if (pending > 0)
        check(cc_engine_drain_events(engine_, 0, events_.data(), pending, &got));
    events_.resize(static_cast<size_t>(got));

    // tree links made during this call -> Point::associated_trees of the roots (cc.cpp:693-694)
    int64_t n_links = 0;
    check(cc_engine_drain_links(engine_, 0, nullptr, 0, &n_links));
    if (n_links > 0)
    {
        link_buf_.resize(static_cast<size_t>(n_links) * 4);
        check(cc_engine_drain_links(engine_, 0, link_buf_.data(), n_links, &got));
        for (int64_t k = 0; k < got; k++)
        {
            const RangeImageIndex a(static_cast<uint16_t>(link_buf_[4 * k + 1]), link_buf_[4 * k + 0] % ring_buffer_max_columns);
            const RangeImageIndex b(static_cast<uint16_t>(link_buf_[4 * k + 3]), link_buf_[4 * k + 2] % ring_buffer_max_columns);
            tree_links_[a].insert(b);
            tree_links_[b].insert(a);
            // Point::associated_trees of both roots (cc.cpp:693-694)
            const size_t Rr = static_cast<size_t>(num_rows_);
            range_image_[static_cast<size_t>(a.column_index) * Rr + a.row_index].associated_trees.insert(b);
            range_image_[static_cast<size_t>(b.column_index) * Rr + b.row_index].associated_trees.insert(a);
        }
    }

    // the columns the events of this call refer to — and only those: the newest column (ground view) and the oldest ones (published) lie a lag of
    // ~100 columns apart, and a call of a few firings gets both from what the engine mirrored with its results (no kernel, no copy: cc_hip.h,
    // cc_engine_read_columns); the mirror is then brought up to date in callback order
    std::vector<std::pair<int64_t, int64_t>> want;
    for (const cc_event& e : events_)
    {
        if (e.type == CC_EV_PUBLISH_COLUMNS && e.b < e.a)
            continue;
        if (e.type == CC_EV_CLUSTER && !(e.d > 20 && finished_cluster_callback_))
            continue;
        want.emplace_back(e.a, e.b);
    }
    fetchRanges(want);

    // member points of the clusters that get a callback: gathered and compacted on the device (cc_engine_gather_cluster_points)
    std::vector<uint32_t> g_cid, g_cnt;
    std::vector<int64_t> g_from, g_to, g_col;
    std::vector<int32_t> g_row;
    if (finished_cluster_callback_)
    {
        for (const cc_event& e : events_)
            if (e.type == CC_EV_CLUSTER && e.d > 20) // cc.cpp:1023
            {
                g_cid.push_back(e.c);
                g_cnt.push_back(e.d);
                g_from.push_back(e.a);
                g_to.push_back(e.b);
            }
        size_t total = 0;
        for (uint32_t c : g_cnt)
            total += c;
        g_col.resize(total);
        g_row.resize(total);
        if (!g_cid.empty())
            check(cc_engine_gather_cluster_points(engine_, 0, static_cast<int64_t>(g_cid.size()), g_cid.data(), g_from.data(),
                                                  g_to.data(), g_cnt.data(), g_col.data(), g_row.data()));
    }
    size_t g_next = 0, g_pos = 0;

    // replay in the order the single-threaded reference invokes its callbacks
    for (const cc_event& e : events_)
    {
        switch (e.type)
        {
            case CC_EV_GROUND_COLUMN:
                if (ring_buffer_start_global_column_index == -1)
                    ring_buffer_start_global_column_index = e.a; // cc.cpp:274-278
                applyColumns(e.a, e.a, STAGE_GROUND);
                if (finished_column_callback_)
                    finished_column_callback_(e.a, e.a, true);
                applyColumns(e.a, e.a, STAGE_ASSOC); // the reference associates the column right after the callback (cc.cpp:622-623)
                break;
            case CC_EV_CLUSTER:
                if (e.d > 20 && finished_cluster_callback_) // cc.cpp:1023
                {
                    const size_t cnt = g_cnt[g_next++];
                    collectClusterPoints(e, g_col.data() + g_pos, g_row.data() + g_pos, cnt);
                    g_pos += cnt;
                    uint64_t min_stamp = std::numeric_limits<uint64_t>::max(), max_stamp = 0;
                    for (const Point& p : cluster_points_)
                    {
                        min_stamp = std::min(min_stamp, p.stamp);
                        max_stamp = std::max(max_stamp, p.stamp);
                    }
                    const uint64_t stamp = config_.clustering.use_last_point_for_cluster_stamp ?
                                               max_stamp :
                                               min_stamp + (max_stamp - min_stamp) / 2; // cc.cpp:1025-1028
                    finished_cluster_callback_(cluster_points_, stamp);
                }
                break;
            case CC_EV_PUBLISH_COLUMNS:
            {
                const int64_t old_start = ring_buffer_start_global_column_index;
                ring_buffer_start_global_column_index = std::max<int64_t>(0, (e.b + 1) - num_columns_); // cc.cpp:1079
                applyColumns(e.a, e.b, STAGE_FULL);
                if (finished_column_callback_)
                    finished_column_callback_(e.a, e.b, false);
                clearMirrorColumns(old_start, ring_buffer_start_global_column_index - 1); // cc.cpp:1091
                break;
            }
            default:
                break;
        }
    }
    // links of trees that finished in a cluster without callback (<= 20 points) or without id: drop what points behind the ring start
    if (!tree_links_.empty() && ring_buffer_start_global_column_index > 0)
        for (auto it = tree_links_.begin(); it != tree_links_.end();)
        {
            const int64_t g = globalColumnOfLocal(it->first.column_index);
            if (g < ring_buffer_start_global_column_index)
                it = tree_links_.erase(it);
            else
                ++it;
        }

    // firings older than anything still in the ring are no longer needed for the pass-through metadata
    int64_t oldest_needed = -1;
    if (ring_buffer_start_global_column_index >= 0)
        for (int64_t g = ring_buffer_start_global_column_index; g < ring_buffer_start_global_column_index + 256 && oldest_needed < 0; g++)
            oldest_needed = col_min_src_[static_cast<size_t>(g % ring_buffer_max_columns)];
    const uint64_t hard_cap = static_cast<uint64_t>(ring_buffer_max_columns) * 16 + 8192;
    while (!firing_log_.empty() &&
           (firing_log_.size() > hard_cap || (oldest_needed >= 0 && static_cast<int64_t>(firing_log_base_) + 8192 < oldest_needed)))
    {
        firing_log_.pop_front();
        firing_log_base_++;
    }
}

} // namespace continuous_clustering
