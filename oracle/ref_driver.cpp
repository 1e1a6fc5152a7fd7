// ref_driver.cpp — TEST INFRASTRUCTURE. A thin extern "C" driver around the REFERENCE's own class
// (continuous_clustering::ContinuousClustering, compiled from /root/reference/src/clustering/continuous_clustering.cpp where it lies),
// exposing the same recording surface as oracle/cc_oracle.cpp so that tests/test_reference_build.py can diff the oracle against the
// reference itself. It only uses the reference's public API (continuous_clustering.hpp:200-251): setConfiguration, reset,
// setTransformRobotFrameFromSensorFrame, addFiring, the two callbacks and the public range_image_ members, driven the way
// src/tools/kitti_demo.cpp:276-313,386-403 drives it (single-threaded mode).
//
// Built ONLY by oracle/build_ref.sh, and only when a real Eigen3 is installed (the reference core needs <Eigen/Geometry>). No stand-in
// headers, ever: without Eigen3 the recipe stops and the oracle stays "parity unpinned". Output: oracle/_ref/libcc_ref.so (git-ignored).
#include <continuous_clustering/clustering/continuous_clustering.hpp>

#include <cstring>
#include <string>
#include <vector>

#include "../include/cc_hip.h"

using namespace continuous_clustering;

namespace
{
struct Snapshot
{
    int64_t gcol;
    std::vector<float> x, y, z, distance, inclination;
    std::vector<double> cont_az, finished_at;
    std::vector<int64_t> cell_gcol, source_firing, root_gcol;
    std::vector<int32_t> root_row, visited;
    std::vector<uint8_t> ground, debug, ignored, finished;
    std::vector<uint64_t> id;
    std::vector<uint32_t> tree_points, width, n_children;
};

struct Ref
{
    ContinuousClustering cc;
    Configuration config;
    int num_rows{0};
    uint64_t seq{0};
    std::vector<cc_event> events;
    std::vector<Snapshot> published;
    int64_t published_base{-1};
    int64_t current_column{-1};
    std::string error;

    void install_callbacks()
    {
        cc.setFinishedColumnCallback(
            [this](int64_t from, int64_t to, bool ground_only)
            {
                cc_event e{};
                e.type = ground_only ? CC_EV_GROUND_COLUMN : CC_EV_PUBLISH_COLUMNS;
                e.a = from;
                e.b = to;
                if (ground_only)
                    current_column = from;
                e.column = current_column;
                events.push_back(e);
                if (!ground_only)
                    for (int64_t g = from; g <= to; g++)
                        snapshot(g);
            });
        cc.setFinishedClusterCallback(
            [this](const std::vector<Point>& pts, uint64_t)
            {
                // (the reference only reports clusters of more than 20 points here, cc.cpp:1023; smaller ones show up as ids in the columns)
                cc_event e{};
                e.type = CC_EV_CLUSTER;
                int64_t lo = INT64_MAX, hi = -1;
                for (const Point& p : pts)
                {
                    lo = std::min(lo, p.global_column_index);
                    hi = std::max(hi, p.global_column_index);
                }
                e.a = lo;
                e.b = hi;
                e.c = pts.empty() ? 0u : (uint32_t) pts[0].id;
                e.d = (uint32_t) pts.size();
                e.column = current_column;
                events.push_back(e);
            });
    }

    void snapshot(int64_t g)
    {
        if (published_base < 0)
            published_base = g;
        Snapshot s;
        s.gcol = g;
        const int lc = (int) (g % cc.ring_buffer_max_columns);
        for (int r = 0; r < num_rows; r++)
        {
            const Point& c = cc.range_image_[(size_t) lc * num_rows + r];
            s.x.push_back(c.xyz.x);
            s.y.push_back(c.xyz.y);
            s.z.push_back(c.xyz.z);
            s.distance.push_back(c.distance);
            s.inclination.push_back(c.inclination_angle);
            s.cont_az.push_back(c.continuous_azimuth_angle);
            s.cell_gcol.push_back(c.global_column_index);
            s.source_firing.push_back(std::isnan(c.distance) ? -1 : (int64_t) c.firing_index);
            s.ground.push_back(c.ground_point_label);
            s.debug.push_back(c.debug_ground_point_label);
            s.ignored.push_back(c.is_ignored ? 1 : 0);
            s.id.push_back(c.id);
            s.finished_at.push_back(c.finished_at_continuous_azimuth_angle);
            s.tree_points.push_back(c.tree_num_points);
            s.width.push_back(c.cluster_width);
            s.n_children.push_back((uint32_t) c.child_points.size());
            s.visited.push_back(c.number_of_visited_neighbors);
            s.finished.push_back(c.belongs_to_finished_cluster ? 1 : 0);
            if (c.tree_root_.column_index >= 0)
            {
                const Point& root = cc.range_image_[(size_t) c.tree_root_.column_index * num_rows + c.tree_root_.row_index];
                s.root_gcol.push_back(root.global_column_index);
                s.root_row.push_back(c.tree_root_.row_index);
            }
            else
            {
                s.root_gcol.push_back(-1);
                s.root_row.push_back(0);
            }
        }
        published.push_back(std::move(s));
    }
};

void to_config(const cc_config& k, Configuration& c)
{
    c.general.is_single_threaded = k.is_single_threaded != 0;
    c.range_image.sensor_is_clockwise = k.sensor_is_clockwise != 0;
    c.range_image.num_columns = k.num_columns;
    c.range_image.supplement_inclination_angle_for_nan_cells = k.supplement_inclination_angle_for_nan_cells != 0;
    auto& g = c.ground_segmentation;
    g.max_slope = k.max_slope;
    g.first_ring_as_ground_max_allowed_z_diff = k.first_ring_as_ground_max_allowed_z_diff;
    g.first_ring_as_ground_min_allowed_z_diff = k.first_ring_as_ground_min_allowed_z_diff;
    g.last_ground_point_slope_higher_than = k.last_ground_point_slope_higher_than;
    g.last_ground_point_distance_smaller_than = k.last_ground_point_distance_smaller_than;
    g.ground_because_close_to_last_certain_ground_max_z_diff = k.ground_because_close_to_last_certain_ground_max_z_diff;
    g.ground_because_close_to_last_certain_ground_max_dist_diff = k.ground_because_close_to_last_certain_ground_max_dist_diff;
    g.obstacle_because_next_certain_obstacle_max_dist_diff = k.obstacle_because_next_certain_obstacle_max_dist_diff;
    g.use_terrain = k.use_terrain != 0;
    g.terrain_max_allowed_z_diff = k.terrain_max_allowed_z_diff;
    g.height_ref_to_maximum_ = k.height_ref_to_maximum_;
    g.height_ref_to_ground_ = k.height_ref_to_ground_;
    g.length_ref_to_front_end_ = k.length_ref_to_front_end_;
    g.length_ref_to_rear_end_ = k.length_ref_to_rear_end_;
    g.width_ref_to_left_mirror_ = k.width_ref_to_left_mirror_;
    g.width_ref_to_right_mirror_ = k.width_ref_to_right_mirror_;
    g.fog_filtering_enabled = k.fog_filtering_enabled != 0;
    g.fog_filtering_intensity_below = (uint8_t) k.fog_filtering_intensity_below;
    g.fog_filtering_distance_below = k.fog_filtering_distance_below;
    g.fog_filtering_inclination_above = k.fog_filtering_inclination_above;
    auto& q = c.clustering;
    q.max_distance = k.max_distance;
    q.max_steps_in_row = k.max_steps_in_row;
    q.max_steps_in_column = k.max_steps_in_column;
    q.stop_after_association_enabled = k.stop_after_association_enabled != 0;
    q.stop_after_association_min_steps = k.stop_after_association_min_steps;
    q.ignore_points_in_chessboard_pattern = k.ignore_points_in_chessboard_pattern != 0;
    q.ignore_points_with_too_big_inclination_angle_diff = k.ignore_points_with_too_big_inclination_angle_diff != 0;
    q.use_last_point_for_cluster_stamp = k.use_last_point_for_cluster_stamp != 0;
    q.cluster_point_trees_every_nth_column = k.cluster_point_trees_every_nth_column;
}

Eigen::Isometry3d iso_from12(const double* m)
{
    Eigen::Isometry3d t = Eigen::Isometry3d::Identity();
    for (int r = 0; r < 3; r++)
    {
        for (int c = 0; c < 3; c++)
            t.linear()(r, c) = m[r * 4 + c];
        t.translation()(r) = m[r * 4 + 3];
    }
    return t;
}
} // namespace

extern "C" {

void* ref_create(const cc_config* cfg, int num_rows)
{
    Ref* h = new Ref();
    cc_config k = *cfg;
    k.is_single_threaded = 1; // the oracle restates the single-threaded order (thread_pool.hpp:58-64)
    to_config(k, h->config);
    h->num_rows = num_rows;
    h->cc.setConfiguration(h->config);
    h->cc.reset(num_rows);
    h->install_callbacks();
    return h;
}

void ref_destroy(void* p)
{
    delete (Ref*) p;
}

void ref_set_robot_from_sensor(void* p, const double* tf12)
{
    ((Ref*) p)->cc.setTransformRobotFrameFromSensorFrame(iso_from12(tf12));
}

// returns CC_OK or the CC_ERR_* class of the std::runtime_error the reference threw
int ref_add_firings(void* p, int64_t n, const float* xyz, const uint8_t* intensity, const double* poses)
{
    Ref& h = *(Ref*) p;
    try
    {
        for (int64_t i = 0; i < n; i++)
        {
            RawPoints::Ptr firing(new RawPoints);
            firing->stamp = h.seq;
            firing->points.resize(h.num_rows);
            for (int r = 0; r < h.num_rows; r++)
            {
                RawPoint& q = firing->points[r];
                q.x = xyz[((size_t) i * h.num_rows + r) * 3 + 0];
                q.y = xyz[((size_t) i * h.num_rows + r) * 3 + 1];
                q.z = xyz[((size_t) i * h.num_rows + r) * 3 + 2];
                q.intensity = intensity[(size_t) i * h.num_rows + r];
                q.firing_index = h.seq; // = the oracle's source_firing
                q.stamp = h.seq;
                q.globally_unique_point_index = (h.seq << 16) | (uint64_t) r;
            }
            h.cc.addFiring(firing, iso_from12(poses + (size_t) i * 12));
            h.seq++;
        }
    }
    catch (const std::runtime_error& e)
    {
        h.error = e.what();
        if (h.error.find("Transform robot frame") != std::string::npos)
            return CC_ERR_NO_ROBOT_TRANSFORM;
        if (h.error.find("not cleared") != std::string::npos)
            return CC_ERR_RING_OVERRUN;
        return CC_ERR_BOOKKEEPING;
    }
    return CC_OK;
}

const char* ref_last_error(void* p)
{
    return ((Ref*) p)->error.c_str();
}

int ref_reset_required(void* p)
{
    return ((Ref*) p)->cc.resetRequired() ? 1 : 0;
}

int64_t ref_num_events(void* p)
{
    return (int64_t) ((Ref*) p)->events.size();
}

void ref_get_events(void* p, cc_event* out)
{
    Ref& h = *(Ref*) p;
    if (!h.events.empty())
        memcpy(out, h.events.data(), h.events.size() * sizeof(cc_event));
}

int64_t ref_published_base(void* p)
{
    return ((Ref*) p)->published_base;
}

int64_t ref_published_count(void* p)
{
    return (int64_t) ((Ref*) p)->published.size();
}

int ref_read_published(void* p, int64_t from, int64_t to, const cc_column_view* v)
{
    Ref& h = *(Ref*) p;
    if (h.published_base < 0 || from < h.published_base || to >= h.published_base + (int64_t) h.published.size() || to < from)
        return CC_ERR_INVALID_ARGUMENT;
    const int R = h.num_rows;
    for (int64_t g = from; g <= to; g++)
    {
        const Snapshot& s = h.published[(size_t) (g - h.published_base)];
        const size_t off = (size_t) (g - from) * R;
        for (int r = 0; r < R; r++)
        {
            if (v->x) v->x[off + r] = s.x[r];
            if (v->y) v->y[off + r] = s.y[r];
            if (v->z) v->z[off + r] = s.z[r];
            if (v->distance) v->distance[off + r] = s.distance[r];
            if (v->inclination_angle) v->inclination_angle[off + r] = s.inclination[r];
            if (v->continuous_azimuth_angle) v->continuous_azimuth_angle[off + r] = s.cont_az[r];
            if (v->global_column_index) v->global_column_index[off + r] = s.cell_gcol[r];
            if (v->source_firing) v->source_firing[off + r] = s.source_firing[r];
            if (v->ground_point_label) v->ground_point_label[off + r] = s.ground[r];
            if (v->debug_ground_point_label) v->debug_ground_point_label[off + r] = s.debug[r];
            if (v->is_ignored) v->is_ignored[off + r] = s.ignored[r];
            if (v->id) v->id[off + r] = s.id[r];
            if (v->tree_root_global_column) v->tree_root_global_column[off + r] = s.root_gcol[r];
            if (v->tree_root_row) v->tree_root_row[off + r] = s.root_row[r];
            if (v->finished_at_continuous_azimuth_angle) v->finished_at_continuous_azimuth_angle[off + r] = s.finished_at[r];
            if (v->tree_num_points) v->tree_num_points[off + r] = s.tree_points[r];
            if (v->cluster_width) v->cluster_width[off + r] = s.width[r];
            if (v->number_of_child_points) v->number_of_child_points[off + r] = s.n_children[r];
            if (v->number_of_visited_neighbors) v->number_of_visited_neighbors[off + r] = s.visited[r];
            if (v->belongs_to_finished_cluster) v->belongs_to_finished_cluster[off + r] = s.finished[r];
        }
    }
    return CC_OK;
}

} // extern "C"
