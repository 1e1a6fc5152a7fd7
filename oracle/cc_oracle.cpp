// cc_oracle.cpp — TEST INFRASTRUCTURE. CPU restatement of the per-column hot path of
// UniBwTAS/continuous_clustering (src/clustering/continuous_clustering.cpp, "cc.cpp" below) in its
// deterministic single-threaded mode (is_single_threaded = true, thread_pool.hpp:58-64: every stage runs
// inline, depth first, inside addFiring).
//
// ** PARITY UNPINNED ** (except libm): the reference core needs Eigen3, which is not in this image and is not
// vendored in /root/reference, and the reference ships no tests, golden vectors or fixtures for this path
// (SURVEY.md 4, 8c). It therefore cannot be compiled here and this restatement could not be checked against
// outputs of the reference itself. What IS pinned: the float atan2/asin it calls are this image's glibc 2.35
// libm (the oracle calls libm directly, like the reference), and the product's device versions are verified
// against that libm exhaustively (oracle/libm_pin.cpp, oracle/libm_pin_full.log).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library. It is the
// checker, never the product: the product (libcc_hip.so) contains no CPU path.
//
// Each function cites the reference lines it follows. The restatement keeps the reference's data structures
// in spirit (array-of-cells ring buffer, child lists, ordered link sets, BFS with the azimuth "visited stamp")
// so that its behaviour — including the quirks listed in SURVEY.md Appendix A — is the reference's, while the
// HIP implementation uses different structures (SoA planes, union-find); agreement between the two is therefore
// a meaningful check.
//
// Rigid transforms: the reference uses Eigen::Isometry3d (3.3.x). Restated here as 3x4 row-major [R|t] doubles with
// the evaluation order  R*p = ((r0*px + r1*py) + r2*pz), then + t  (what Eigen's unrolled 3x3 * 3x1 product
// followed by "+ translation()" computes without FMA); inverse = (R^T, -(R^T) t); norm = sqrt((x^2+y^2)+z^2).
// For identity or pure-translation poses every one of these is exact, so the order is immaterial.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <list>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>
#include <thread>
#include <mutex>
#include <atomic>

#include "../include/cc_hip.h"

namespace
{

struct Iso
{
    double r[9];
    double t[3];
};

inline Iso iso_from12(const double* m)
{
    Iso o;
    for (int i = 0; i < 3; i++)
    {
        for (int j = 0; j < 3; j++)
            o.r[i * 3 + j] = m[i * 4 + j];
        o.t[i] = m[i * 4 + 3];
    }
    return o;
}

inline void iso_apply(const Iso& a, const double p[3], double out[3])
{
    for (int i = 0; i < 3; i++)
        out[i] = ((a.r[i * 3 + 0] * p[0] + a.r[i * 3 + 1] * p[1]) + a.r[i * 3 + 2] * p[2]) + a.t[i];
}

inline Iso iso_inverse(const Iso& a)
{
    Iso o;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            o.r[i * 3 + j] = a.r[j * 3 + i];
    for (int i = 0; i < 3; i++)
        o.t[i] = ((-o.r[i * 3 + 0]) * a.t[0] + (-o.r[i * 3 + 1]) * a.t[1]) + (-o.r[i * 3 + 2]) * a.t[2];
    return o;
}

inline Iso iso_mul(const Iso& a, const Iso& b)
{
    Iso o;
    for (int i = 0; i < 3; i++)
    {
        for (int j = 0; j < 3; j++)
            o.r[i * 3 + j] =
                (a.r[i * 3 + 0] * b.r[0 * 3 + j] + a.r[i * 3 + 1] * b.r[1 * 3 + j]) + a.r[i * 3 + 2] * b.r[2 * 3 + j];
        o.t[i] = ((a.r[i * 3 + 0] * b.t[0] + a.r[i * 3 + 1] * b.t[1]) + a.r[i * 3 + 2] * b.t[2]) + a.t[i];
    }
    return o;
}

// (row, local column) handle of a cell; ordering = RangeImageIndex::operator< (cc.hpp:107-110): row first.
struct Ref
{
    int64_t col{-1};
    uint16_t row{0};
    bool operator<(const Ref& o) const
    {
        return row < o.row || (row == o.row && col < o.col);
    }
    bool operator==(const Ref& o) const
    {
        return row == o.row && col == o.col;
    }
    bool operator!=(const Ref& o) const
    {
        return !(*this == o);
    }
};

// One range-image cell: the algorithm-owned fields of struct Point (cc.hpp:126-161).
struct Cell
{
    float x, y, z;
    float distance, azimuth, inclination;
    double cont_az;
    int64_t gcol;
    int32_t lcol, row;
    int64_t source_firing;
    uint8_t intensity;
    uint8_t ground, debug;
    bool ignored;
    double finished_at;
    std::list<Ref> children;
    std::set<Ref> links;
    Ref root;
    Ref parent; // the cell whose child list holds this point (bookkeeping of the checker: the reference only keeps the child lists)
    uint32_t tree_points, width;
    uint64_t tree_id, id;
    double visited_at;
    bool finished;
    int visited_neighbors;
};

struct ColumnSnapshot
{
    int64_t gcol;
    std::vector<float> x, y, z, distance, inclination;
    std::vector<double> cont_az;
    std::vector<int64_t> cell_gcol, source_firing, root_gcol;
    std::vector<int32_t> root_row;
    std::vector<uint8_t> ground, debug, ignored;
    std::vector<uint64_t> id;
    std::vector<double> finished_at;
    std::vector<uint32_t> tree_points, width, n_children;
    std::vector<int32_t> visited, parent_row;
    std::vector<uint8_t> finished;
    std::vector<int64_t> parent_gcol;
};

struct Oracle
{
    cc_config cfg{};
    int num_rows{-1}, num_columns{0}, ring_cols{0};
    std::vector<Cell> img;
    int64_t ring_start{-1}, ring_end{-1};

    // srig_* (cc.hpp:257-262)
    float az_width{0};
    int64_t prev_rearmost{0}, prev_foremost{-1}, first_unfinished{-1};
    double sensor_pos_d[3]{0, 0, 0};
    bool reset_required{false};
    // sgps_* (cc.hpp:265-266)
    float sensor_pos_f[3]{0, 0, 0};
    bool has_robot_tf{false};
    Iso robot_from_sensor{};
    // sc_* (cc.hpp:269-275)
    float max_distance_squared{0.7f * 0.7f};
    int64_t first_unpublished{-1};
    std::list<int64_t> min_required_list;
    std::list<Ref> unfinished;
    uint64_t cluster_counter{1};
    std::vector<float> incl_steps;

    // recording
    bool record{true};
    std::vector<cc_event> events;
    std::vector<ColumnSnapshot> published; // snapshots taken inside the cluster-view column callback
    int64_t published_base{-1};            // gcol of published[0]
    std::vector<std::vector<std::pair<int64_t, int32_t>>> cluster_members; // per CC_EV_CLUSTER event, in event order (record only)
    int64_t keep_tail{0};                  // > 0: only the most recent snapshots are kept (long verification runs of bench.py)
    uint64_t firings_consumed{0}, cells_published{0}, clusters_finished{0};
    uint64_t exceed_one_rotation{0};
    // ---- BASELINE.md mode B (orc_time_firings_pipeline below): the stages handed from thread to thread instead of called depth first
    struct Pipe;
    Pipe* pipe{nullptr};
    float seg_pos[3]{0, 0, 0}; // sgps_sensor_position of the column being segmented (pipeline: owned by the segmentation thread)
    bool ring_inited{false};   // (what `ring_start == -1` tells insert_firing, without reading a field the publishing thread writes)
    void pipe_push_segment(int64_t gcol, const Iso& pose);
    void pipe_push_associate(int64_t gcol);
    uint64_t max_unfinished{0};
    std::string error;

    Cell& at(int64_t lcol, int row)
    {
        return img[(size_t) lcol * num_rows + row];
    }

    // ---- cc.cpp:1094-1145 clearColumns ------------------------------------------------------------------
    void clear_columns(int64_t from, int64_t to)
    {
        if (to < from)
            return;
        const float fn = std::nanf("");
        for (int64_t g = from; g <= to; g++)
        {
            int lc = (int) (g % ring_cols);
            for (int r = 0; r < num_rows; r++)
            {
                Cell& c = at(lc, r);
                c.x = c.y = c.z = fn;
                c.distance = c.azimuth = c.inclination = fn;
                c.cont_az = std::nan("");
                c.gcol = -1;
                c.lcol = -1;
                c.row = -1;
                c.intensity = 0;
                c.source_firing = -1;
                c.ground = CC_GP_UNKNOWN;
                c.debug = CC_DBG_WHITE;
                c.ignored = false;
                c.finished_at = 0.f;
                c.children.clear();
                c.links.clear();
                c.root.row = 0;
                c.root.col = -1;
                c.parent.row = 0;
                c.parent.col = -1;
                c.tree_points = 0;
                c.width = 0;
                c.tree_id = 0;
                c.id = 0;
                c.visited_at = -1.;
                c.finished = false;
                c.visited_neighbors = 0;
            }
        }
    }

    // ---- cc.cpp:11-64 reset ---------------------------------------------------------------------------
    void reset(int rows)
    {
        num_columns = cfg.num_columns;
        num_rows = rows;
        az_width = static_cast<float>((2 * M_PI)) / static_cast<float>(num_columns);
        ring_cols = num_columns * 10;
        img.assign((size_t) ring_cols * rows, Cell{});
        clear_columns(0, ring_cols - 1);
        ring_start = -1;
        ring_end = -1;
        ring_inited = false;
        prev_rearmost = 0;
        prev_foremost = -1;
        first_unfinished = -1;
        reset_required = false;
        has_robot_tf = false;
        first_unpublished = -1;
        min_required_list.clear();
        unfinished.clear();
        cluster_counter = 1;
        // std::vector::resize(n, v) keeps existing elements (cc.cpp:46) — a fresh oracle starts all-NaN.
        incl_steps.resize(rows, std::nanf(""));
        events.clear();
        published.clear();
        cluster_members.clear();
        published_base = -1;
        firings_consumed = cells_published = clusters_finished = 0;
    }

    // ---- cc.cpp:66-81 setConfiguration ----------------------------------------------------------------
    void set_config(const cc_config& c)
    {
        if ((cfg.is_single_threaded != 0) != (c.is_single_threaded != 0))
            reset_required = true;
        if ((cfg.sensor_is_clockwise != 0) != (c.sensor_is_clockwise != 0))
            reset_required = true;
        if (cfg.num_columns != c.num_columns)
            reset_required = true;
        cfg = c;
        max_distance_squared = cfg.max_distance * cfg.max_distance;
    }

    std::mutex emit_mutex; // (pipeline mode with recording on: two stage threads append events)
    void emit(int type, int64_t a, int64_t b, uint32_t c, uint32_t d, int64_t column)
    {
        if (!record)
            return;
        std::unique_lock<std::mutex> lk(emit_mutex, std::defer_lock);
        if (pipe)
            lk.lock();
        cc_event e{};
        e.type = type;
        e.stream = 0;
        e.a = a;
        e.b = b;
        e.c = c;
        e.d = d;
        e.column = column;
        events.push_back(e);
    }

    // ---- cc.cpp:105-292 insertFiringIntoRangeImage ----------------------------------------------------
    void insert_firing(const float* xyz, const uint8_t* intensity, const Iso& odom_from_sensor, int64_t firing_seq)
    {
        for (int i = 0; i < 3; i++)
        {
            sensor_pos_d[i] = odom_from_sensor.t[i];
            sensor_pos_f[i] = static_cast<float>(sensor_pos_d[i]);
        }
        int64_t foremost = -1, rearmost = -1;
        int64_t prev_rot = prev_rearmost / num_columns;

        for (int row = 0; row < num_rows; row++)
        {
            double p[3] = {xyz[row * 3 + 0], xyz[row * 3 + 1], xyz[row * 3 + 2]};
            if (std::isnan(p[0]))
                continue;
            double p_odom[3];
            iso_apply(odom_from_sensor, p, p_odom);
            double rel[3] = {p_odom[0] - sensor_pos_d[0], p_odom[1] - sensor_pos_d[1], p_odom[2] - sensor_pos_d[2]};

            float azimuth = std::atan2(static_cast<float>(p[1]), static_cast<float>(p[0])); // sensor frame, :142
            float inc_az = cfg.sensor_is_clockwise ? -azimuth + static_cast<float>(M_PI) : azimuth + static_cast<float>(M_PI);

            int col_in_rot = static_cast<int>(inc_az / az_width);
            int64_t gcol = prev_rot * num_columns + col_in_rot;

            int prev_col_in_rot = static_cast<int>(prev_rearmost % num_columns);
            int column_diff = col_in_rot - prev_col_in_rot;
            int half = num_columns / 2;
            int rot_offset = 0;
            if (column_diff < -half)
            {
                gcol += num_columns;
                rot_offset = 1;
            }
            else if (prev_rearmost > 0 && column_diff > half)
            {
                gcol -= num_columns;
                rot_offset = -1;
            }
            if (gcol < 0)
            {
                // The reference would index range_image_ with a negative local column here (undefined behaviour,
                // cc.cpp:178-181). Both oracle and product drop the return and flag the stream.
                if (error.empty())
                    error = "negative global column index";
                continue;
            }
            int lcol = static_cast<int>(gcol % ring_cols);
            Cell* cell = &at(lcol, row);

            double cont_az = (2 * M_PI) * static_cast<double>(prev_rot + rot_offset) + inc_az;

            float distance = static_cast<float>(std::sqrt((rel[0] * rel[0] + rel[1] * rel[1]) + rel[2] * rel[2]));
            if (!std::isnan(cell->distance) && !std::isnan(distance))
            {
                int next = lcol + 1;
                if (next >= ring_cols)
                    next -= ring_cols;
                Cell* nc = &at(next, row);
                if (std::isnan(nc->distance))
                {
                    cell = nc;
                    lcol = next;
                    gcol++;
                }
            }
            if (!std::isnan(cell->distance) && (std::isnan(distance) || distance >= cell->distance))
                continue;

            bool too_far_behind = first_unfinished >= 0 && gcol < first_unfinished;
            if (!too_far_behind)
            {
                cell->x = static_cast<float>(p_odom[0]);
                cell->y = static_cast<float>(p_odom[1]);
                cell->z = static_cast<float>(p_odom[2]);
                cell->intensity = intensity[row];
                cell->source_firing = firing_seq;
                cell->distance = distance;
                cell->azimuth = azimuth;
                cell->inclination = std::asin(static_cast<float>(rel[2]) / cell->distance);
                cell->cont_az = cont_az;
                cell->gcol = gcol;
                cell->lcol = lcol;
                cell->row = row;
            }
            if (rearmost < 0 || gcol < rearmost)
                rearmost = gcol;
            if (foremost < 0 || gcol > foremost)
                foremost = gcol;
        }

        if (rearmost >= 0 && foremost >= 0)
        {
            if ((foremost - rearmost) > num_columns / 2)
            {
                reset_required = true; // cc.cpp:252-261 (the reference also prints one line)
                return;
            }
            if (rearmost > prev_rearmost)
                prev_rearmost = rearmost;
            if (foremost > prev_foremost)
                prev_foremost = foremost;
        }
        if (prev_foremost < 0)
            return;
        if (!ring_inited)
        {
            ring_inited = true;
            ring_start = prev_rearmost;
            first_unpublished = prev_rearmost;
        }
        if (prev_foremost > ring_end)
            ring_end = prev_foremost;
        if (first_unfinished == -1)
            first_unfinished = prev_rearmost;
        while (first_unfinished < prev_rearmost)
        {
            if (pipe)
                pipe_push_segment(first_unfinished++, odom_from_sensor); // (the segmentation thread takes it from here)
            else
            {
                for (int i = 0; i < 3; i++)
                    seg_pos[i] = sensor_pos_f[i];
                segment_column(first_unfinished++, odom_from_sensor);
            }
        }
    }

    static inline float len2(float a, float b)
    {
        return std::sqrt(a * a + b * b);
    }

    // ---- cc.cpp:294-624 performGroundPointSegmentationForColumn ---------------------------------------
    void segment_column(int64_t gcol, const Iso& odom_from_sensor)
    {
        int lc = static_cast<int>(gcol % ring_cols);
        if (!has_robot_tf)
            throw std::runtime_error("Transform robot frame from sensor frame was not set yet!");
        Iso ego_from_odom = iso_mul(robot_from_sensor, iso_inverse(odom_from_sensor));
        float height_sensor_to_ground = -static_cast<float>(robot_from_sensor.t[2]) + cfg.height_ref_to_ground_;

        bool first_obstacle_detected = false;
        bool first_point_found = false;
        float last_ground[3] = {0, 0, height_sensor_to_ground};
        float prev_pos[3] = {0, 0, 0};
        uint8_t previous_label = 0;
        float incl_prev_laser = 0;

        for (int row = num_rows - 1; row >= 0; row--)
        {
            Cell& c = at(lc, row);
            int64_t g = c.gcol;
            if (g != gcol && g != -1)
                throw std::runtime_error("This column is not cleared. Probably this means the ring buffer is full or there "
                                         "is some other issue with clearing (not cleared at all or written after clearing): " +
                                         std::to_string(g) + ", " + std::to_string(gcol) + ", " + std::to_string(ring_cols));
            c.gcol = gcol;
            c.lcol = lc;

            float incl_cur = c.inclination;
            float diff = incl_cur - incl_prev_laser;
            if (!std::isnan(diff))
                incl_steps[row] = diff;
            incl_prev_laser = incl_cur;

            if (std::isnan(c.distance))
            {
                if (cfg.supplement_inclination_angle_for_nan_cells && row < num_rows - 1)
                    c.inclination = at(lc, row + 1).inclination + incl_steps[row];
                c.cont_az = (static_cast<double>(gcol) + 0.5) * az_width;
                continue;
            }

            if (cfg.fog_filtering_enabled && c.intensity < static_cast<uint8_t>(cfg.fog_filtering_intensity_below) &&
                c.distance < cfg.fog_filtering_distance_below && c.inclination > cfg.fog_filtering_inclination_above)
            {
                c.ground = CC_GP_FOG;
                c.debug = CC_DBG_LIGHTGRAY;
                continue;
            }

            double pd[3] = {c.x, c.y, c.z}, pe[3];
            iso_apply(ego_from_odom, pd, pe);
            if (pe[0] < cfg.length_ref_to_front_end_ && pe[0] > cfg.length_ref_to_rear_end_ &&
                pe[1] < cfg.width_ref_to_left_mirror_ && pe[1] > cfg.width_ref_to_right_mirror_ &&
                pe[2] < cfg.height_ref_to_maximum_ && pe[2] > cfg.height_ref_to_ground_)
            {
                c.ground = CC_GP_EGO_VEHICLE;
                c.debug = CC_DBG_VIOLET;
                continue;
            }

            float cur[3] = {c.x - seg_pos[0], c.y - seg_pos[1], c.z - seg_pos[2]};

            if (!first_point_found)
            {
                first_point_found = true;
                float h = cur[2] - height_sensor_to_ground;
                if (h > cfg.first_ring_as_ground_min_allowed_z_diff && h < cfg.first_ring_as_ground_max_allowed_z_diff)
                {
                    c.ground = CC_GP_GROUND;
                    c.debug = CC_DBG_GRAY;
                    last_ground[0] = cur[0];
                    last_ground[1] = cur[1];
                    last_ground[2] = cur[2];
                    first_obstacle_detected = false;
                }
                else
                {
                    c.ground = CC_GP_OBSTACLE;
                    c.debug = CC_DBG_ORANGE;
                    first_obstacle_detected = true;
                }
                prev_pos[0] = cur[0];
                prev_pos[1] = cur[1];
                prev_pos[2] = cur[2];
                previous_label = c.debug;
                continue;
            }

            // azimuth-plane coordinates: (||xy||, z)   cc.hpp:229-232, general.hpp:15-18
            float cur2x = len2(cur[0], cur[1]), cur2y = cur[2];
            float prv2x = len2(prev_pos[0], prev_pos[1]), prv2y = prev_pos[2];
            float p2c_x = cur2x - prv2x, p2c_y = cur2y - prv2y;
            float slope_to_prev = p2c_y / p2c_x;
            bool flat_prev = std::abs(slope_to_prev) < cfg.max_slope && p2c_x > 0;
            flat_prev = flat_prev && (!cfg.use_terrain || p2c_x < 5);

            float lg2x = len2(last_ground[0], last_ground[1]), lg2y = last_ground[2];
            float l2c_x = cur2x - lg2x, l2c_y = cur2y - lg2y;
            float slope_to_lg = l2c_y / l2c_x;
            bool flat_lg = std::abs(slope_to_lg) < cfg.max_slope && l2c_x > 0;

            if (!first_obstacle_detected && flat_prev)
            {
                c.ground = CC_GP_GROUND;
                c.debug = CC_DBG_GREEN;
            }
            else if (!cfg.use_terrain) // terrain branch is commented out in the reference (cc.cpp:455-489)
            {
                if (first_obstacle_detected && flat_prev && flat_lg)
                {
                    c.ground = CC_GP_GROUND;
                    c.debug = CC_DBG_YELLOWGREEN;
                }
                else if (std::abs(l2c_x) < cfg.ground_because_close_to_last_certain_ground_max_dist_diff &&
                         std::abs(l2c_y) < cfg.ground_because_close_to_last_certain_ground_max_z_diff)
                {
                    c.ground = CC_GP_GROUND;
                    c.debug = CC_DBG_YELLOW;
                }
            }

            if (c.ground != CC_GP_GROUND)
            {
                c.ground = CC_GP_OBSTACLE;
                c.debug = CC_DBG_RED;
                int below = row + 1;
                while (below < num_rows)
                {
                    Cell& b = at(lc, below);
                    float bx = len2(b.x - seg_pos[0], b.y - seg_pos[1]);
                    if (b.debug == CC_DBG_YELLOW ||
                        (b.ground == CC_GP_GROUND &&
                         std::abs(cur2x - bx) < cfg.obstacle_because_next_certain_obstacle_max_dist_diff))
                    {
                        if (b.ground == CC_GP_GROUND)
                        {
                            b.ground = CC_GP_OBSTACLE;
                            b.debug = CC_DBG_DARKRED;
                        }
                        below++;
                    }
                    else
                        break;
                }
            }

            first_obstacle_detected |= c.ground == CC_GP_OBSTACLE;

            if (c.debug == CC_DBG_GREEN || c.debug == CC_DBG_YELLOWGREEN)
            {
                if (slope_to_prev > cfg.last_ground_point_slope_higher_than &&
                    std::abs(p2c_x) < cfg.last_ground_point_distance_smaller_than && previous_label != CC_DBG_YELLOW)
                {
                    last_ground[0] = cur[0];
                    last_ground[1] = cur[1];
                    last_ground[2] = cur[2];
                }
            }
            prev_pos[0] = cur[0];
            prev_pos[1] = cur[1];
            prev_pos[2] = cur[2];
            previous_label = c.debug;
        }

        // second loop: ignore flags (cc.cpp:567-616)
        for (int row = num_rows - 1; row >= 0; row--)
        {
            Cell& c = at(lc, row);
            c.ignored = false;
            if (std::isnan(c.distance))
            {
                c.ignored = true;
                continue;
            }
            if (c.ground != CC_GP_OBSTACLE)
            {
                c.ignored = true;
                continue;
            }
            if (c.distance < 1. * cfg.max_distance)
            {
                c.ignored = true;
                continue;
            }
            if (cfg.ignore_points_with_too_big_inclination_angle_diff && row < (num_rows - 1) &&
                std::atan2(cfg.max_distance, c.distance) < incl_steps[row])
            {
                c.ignored = true;
                continue;
            }
            if (cfg.ignore_points_in_chessboard_pattern)
            {
                bool column_even = c.gcol % 2 == 0;
                bool row_even = row % 2 == 0;
                if ((column_even && !row_even) || (!column_even && row_even))
                {
                    c.ignored = true;
                    continue;
                }
            }
        }

        emit(CC_EV_GROUND_COLUMN, gcol, gcol, 0, 0, gcol);
        if (pipe)
            pipe_push_associate(gcol); // (the association thread takes it from here)
        else
            associate_column(gcol);
    }

    // ---- cc.cpp:638-641 -------------------------------------------------------------------------------
    bool close_enough(const Cell& a, const Cell& b) const
    {
        float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
        return dx * dx + dy * dy + dz * dz < max_distance_squared;
    }

    // ---- cc.cpp:643-673 -------------------------------------------------------------------------------
    void attach_to_tree(Cell& p, Cell& other, float max_angle_diff)
    {
        Cell& root = at(other.root.col, other.root.row);
        uint32_t new_width = static_cast<uint32_t>(p.gcol - root.gcol + 1);
        bool smaller_than_rotation = new_width <= static_cast<uint32_t>(num_columns);
        if (smaller_than_rotation && !root.finished)
        {
            p.root = other.root;
            p.tree_id = root.gcol * num_rows + root.row;
            other.children.push_back(Ref{p.lcol, static_cast<uint16_t>(p.row)});
            p.parent = Ref{other.lcol, static_cast<uint16_t>(other.row)};
            root.width = new_width;
            root.finished_at = std::max(root.finished_at, p.cont_az + max_angle_diff);
            root.tree_points++;
        }
    }

    // ---- cc.cpp:675-696 -------------------------------------------------------------------------------
    void link_trees(const Cell& p, const Cell& other)
    {
        Cell& ra = at(p.root.col, p.root.row);
        Cell& rb = at(other.root.col, other.root.row);
        if (!ra.finished && !rb.finished)
        {
            ra.links.insert(other.root);
            rb.links.insert(p.root);
        }
    }

    // ---- cc.cpp:698-771 traverseFieldOfView -----------------------------------------------------------
    void traverse(Cell& p, float mad, int first_local_col)
    {
        int steps_back_needed = static_cast<int>(std::ceil(mad / az_width));
        steps_back_needed = std::min(steps_back_needed, (int) cfg.max_steps_in_row);
        int64_t oc = p.lcol;
        for (int sb = 0; sb <= steps_back_needed; sb++)
        {
            for (int dir = -1; dir <= 1; dir += 2)
            {
                if (dir == 1 && sb == 0)
                    continue;
                int sv = (dir == 1 || sb == 0) ? 1 : 0;
                int orow = (dir == 1 || sb == 0) ? p.row + dir : p.row;
                while (orow >= 0 && orow < num_rows && sv <= cfg.max_steps_in_column)
                {
                    Cell& o = at(oc, orow);
                    p.visited_neighbors += 1;
                    if (std::abs(o.inclination - p.inclination) > mad)
                        break;
                    if (!o.ignored && (p.root.col == 0 || o.root != p.root))
                    {
                        if (close_enough(p, o))
                        {
                            if (p.root.col == -1)
                                attach_to_tree(p, o, mad);
                            else
                                link_trees(p, o);
                        }
                    }
                    if (p.root.col != -1 && cfg.stop_after_association_enabled && sv >= cfg.stop_after_association_min_steps)
                        break;
                    orow += dir;
                    sv++;
                }
            }
            if (p.root.col != -1 && cfg.stop_after_association_enabled && sb >= cfg.stop_after_association_min_steps)
                break;
            if (oc == first_local_col)
                break;
            oc--;
            if (oc < 0)
                oc += ring_cols;
        }
    }

    // ---- cc.cpp:773-835 associatePointsInColumn -------------------------------------------------------
    void associate_column(int64_t gcol)
    {
        std::list<Ref> new_trees;
        double col_min_az = std::numeric_limits<double>::max();
        int first_local = static_cast<int>(first_unpublished % ring_cols);
        int lc = static_cast<int>(gcol % ring_cols);
        for (int row = 0; row < num_rows; row++)
        {
            Cell& p = at(lc, row);
            if (p.cont_az < col_min_az)
                col_min_az = p.cont_az;
            if (p.ignored)
                continue;
            Ref self{lc, static_cast<uint16_t>(row)};
            float mad = std::asin(cfg.max_distance / p.distance);
            traverse(p, mad, first_local);
            if (p.root.col == -1)
            {
                p.root = self;
                p.tree_id = p.gcol * num_rows + p.row;
                p.finished_at = p.cont_az + mad;
                p.width = 1;
                p.tree_points = 1;
                new_trees.push_back(self);
            }
        }
        combine_trees(gcol, new_trees, col_min_az);
    }

    // ---- cc.cpp:837-974 findFinishedTreesAndAssignSameId ----------------------------------------------
    void combine_trees(int64_t gcol, std::list<Ref>& new_trees, double col_min_az)
    {
        unfinished.splice(unfinished.end(), new_trees);
        if (unfinished.size() > max_unfinished)
            max_unfinished = unfinished.size();
        if (gcol % cfg.cluster_point_trees_every_nth_column != 0)
            return;

        std::list<std::list<Ref>> trees_per_cluster;
        std::list<uint64_t> cluster_ids;
        std::list<Ref> collected, to_visit;
        for (Ref& start : unfinished)
        {
            Cell& sr = at(start.col, start.row);
            if (sr.visited_at == col_min_az)
                continue;
            collected.clear();
            to_visit.clear();
            to_visit.push_back(start);
            int64_t min_col = std::numeric_limits<int64_t>::max();
            int64_t max_col = 0;
            uint32_t num_points = 0;
            bool any_unfinished = false;
            while (!to_visit.empty())
            {
                Ref cur = to_visit.front();
                to_visit.pop_front();
                Cell& cr = at(cur.col, cur.row);
                if (cr.finished)
                    continue;
                min_col = std::min(min_col, cr.gcol);
                max_col = std::max(max_col, cr.gcol + cr.width);
                if (cr.finished_at > col_min_az)
                    any_unfinished = true;
                if (cr.visited_at == col_min_az)
                    continue;
                cr.visited_at = col_min_az;
                collected.push_back(cur);
                num_points += cr.tree_points;
                for (const Ref& nb : cr.links)
                {
                    Cell& nr = at(nb.col, nb.row);
                    if (nr.visited_at != col_min_az)
                        to_visit.push_back(nb);
                }
            }
            bool exceeds = false;
            if (max_col - min_col >= num_columns)
            {
                exceed_one_rotation++; // the reference prints "Found a cluster exceeding one rotation" (cc.cpp:916)
                exceeds = true;
            }
            if ((collected.empty() || any_unfinished) && !exceeds)
                continue;
            for (const Ref& t : collected)
                at(t.col, t.row).finished = true;
            if (num_points > 5)
            {
                // extent of the cluster for the event record
                trees_per_cluster.push_back(std::move(collected));
                cluster_ids.push_back(cluster_counter++);
                collected = std::list<Ref>();
            }
        }

        int64_t min_required = std::numeric_limits<int64_t>::max();
        for (auto it = unfinished.begin(); it != unfinished.end();)
        {
            Cell& r = at(it->col, it->row);
            if (r.gcol < min_required)
                min_required = r.gcol;
            if (r.finished)
                it = unfinished.erase(it);
            else
                ++it;
        }
        if (min_required == std::numeric_limits<int64_t>::max())
            min_required = gcol + 1;
        min_required_list.push_back(min_required);
        publish(gcol, min_required, cluster_ids, trees_per_cluster);
    }

    // ---- cc.cpp:976-1092 collectPointsForCusterAndPublish ---------------------------------------------
    void publish(int64_t gcol, int64_t min_required, std::list<uint64_t>& ids, std::list<std::list<Ref>>& trees)
    {
        auto it_trees = trees.begin();
        for (uint64_t cid : ids)
        {
            uint32_t n = 0;
            int64_t cmin = std::numeric_limits<int64_t>::max(), cmax = -1;
            std::list<Ref> q;
            std::vector<std::pair<int64_t, int32_t>> members; // (global column, row) in the order cluster_points is filled
            for (const Ref& t : *it_trees)
            {
                q.clear();
                q.push_back(t);
                while (!q.empty())
                {
                    Ref cur = q.front();
                    q.pop_front();
                    Cell& c = at(cur.col, cur.row);
                    c.id = cid;
                    if (record)
                        members.emplace_back(c.gcol, (int32_t) cur.row);
                    n++;
                    cmin = std::min(cmin, c.gcol);
                    cmax = std::max(cmax, c.gcol);
                    for (const Ref& ch : c.children)
                        q.push_back(ch);
                }
            }
            clusters_finished++;
            if (record)
                cluster_members.push_back(std::move(members));
            emit(CC_EV_CLUSTER, cmin, cmax, (uint32_t) cid, n, gcol);
            ++it_trees;
        }

        auto pos = std::lower_bound(min_required_list.begin(), min_required_list.end(), min_required);
        if (pos != min_required_list.end() && *pos == min_required)
            min_required_list.erase(pos);
        else
            throw std::runtime_error("The minimum unprocessed column index is not available! This is a bug!");
        int64_t ring_start_old = ring_start;
        int64_t unpub_old = first_unpublished;
        if (!min_required_list.empty())
            first_unpublished = min_required_list.front();
        else
            first_unpublished = min_required;
        if (first_unpublished < unpub_old)
            throw std::runtime_error("This shouldn't happen, ring buffer is not allowed to increase at the front: " +
                                     std::to_string(first_unpublished) + ", " + std::to_string(unpub_old));
        ring_start = std::max((int64_t) 0, first_unpublished - num_columns);
        // cluster-view column callback for [unpub_old, first_unpublished-1]
        emit(CC_EV_PUBLISH_COLUMNS, unpub_old, first_unpublished - 1, 0, 0, gcol);
        if (first_unpublished > unpub_old)
            cells_published += (uint64_t) (first_unpublished - unpub_old) * num_rows;
        if (record)
            for (int64_t g = unpub_old; g < first_unpublished; g++)
                snapshot(g);
        clear_columns(ring_start_old, ring_start - 1);
    }

    void snapshot(int64_t g)
    {
        if (published_base < 0)
            published_base = g;
        ColumnSnapshot s;
        s.gcol = g;
        int lc = (int) (g % ring_cols);
        for (int r = 0; r < num_rows; r++)
        {
            Cell& c = at(lc, r);
            s.x.push_back(c.x);
            s.y.push_back(c.y);
            s.z.push_back(c.z);
            s.distance.push_back(c.distance);
            s.inclination.push_back(c.inclination);
            s.cont_az.push_back(c.cont_az);
            s.cell_gcol.push_back(c.gcol);
            s.source_firing.push_back(c.source_firing);
            s.ground.push_back(c.ground);
            s.debug.push_back(c.debug);
            s.ignored.push_back(c.ignored ? 1 : 0);
            s.id.push_back(c.id);
            s.finished_at.push_back(c.finished_at);
            s.tree_points.push_back(c.tree_points);
            s.width.push_back(c.width);
            s.n_children.push_back((uint32_t) c.children.size());
            s.visited.push_back(c.visited_neighbors);
            s.finished.push_back(c.finished ? 1 : 0);
            if (c.parent.col >= 0)
            {
                s.parent_gcol.push_back(at(c.parent.col, c.parent.row).gcol);
                s.parent_row.push_back(c.parent.row);
            }
            else
            {
                s.parent_gcol.push_back(-1);
                s.parent_row.push_back(0);
            }
            if (c.root.col >= 0)
            {
                s.root_gcol.push_back(at(c.root.col, c.root.row).gcol);
                s.root_row.push_back(c.root.row);
            }
            else
            {
                s.root_gcol.push_back(-1);
                s.root_row.push_back(0);
            }
        }
        published.push_back(std::move(s));
        if (keep_tail > 0 && (int64_t) published.size() > 2 * keep_tail)
        {
            const int64_t drop = (int64_t) published.size() - keep_tail;
            published.erase(published.begin(), published.begin() + drop);
            published_base += drop;
        }
    }

    // ---- cc.cpp:88-93 addFiring -----------------------------------------------------------------------
    void add_firing(const float* xyz, const uint8_t* intensity, const double* pose12)
    {
        insert_firing(xyz, intensity, iso_from12(pose12), (int64_t) firings_consumed);
        firings_consumed++;
    }
};

// ---- BASELINE.md 3, mode B: "reference pipeline" ---------------------------------------------------------------------------------------
// The reference's multi-threaded mode (cc.cpp:49-63) hands every column from stage to stage through thread pools: insertion (1 thread),
// segmentation (1), association (1), tree combination (1), publishing (3). Its stages share the range image, the tree lists and
// sc_first_unpublished without further synchronisation (the association of column c + 1 may run while column c's trees are combined). This
// restatement keeps the single-threaded data structures, so it runs the part of that pipeline that is free of data races on them: THREE
// threads — insertion (the caller), segmentation, association + combination + publishing — connected by bounded single-producer
// single-consumer queues (the producer waits when a queue is full: back-pressure, so the ring never overruns, cc.cpp:337). Results are the
// single-threaded ones (every column still passes the stages in order). What it measures: how much of the reference's per-column work
// overlaps when insertion (35 % of the single-threaded time), segmentation (12 %) and the rest (~ 50 %) run concurrently.
struct Oracle::Pipe
{
    struct SegJob
    {
        int64_t gcol;
        Iso pose;
        float pos[3];
    };
    static constexpr size_t CAP = 1024; // columns in flight per queue (a small fraction of the 10-rotation ring)
    std::vector<SegJob> seg{CAP};
    std::vector<int64_t> assoc = std::vector<int64_t>(CAP);
    std::atomic<uint64_t> seg_head{0}, seg_tail{0}, assoc_head{0}, assoc_tail{0};
    std::atomic<bool> done_insert{false}, done_segment{false}, failed{false};
    std::string error;
};

void Oracle::pipe_push_segment(int64_t gcol, const Iso& pose)
{
    Pipe& q = *pipe;
    const uint64_t t = q.seg_tail.load(std::memory_order_relaxed);
    while (t - q.seg_head.load(std::memory_order_acquire) >= Pipe::CAP && !q.failed.load())
        std::this_thread::yield();
    Pipe::SegJob& j = q.seg[t % Pipe::CAP];
    j.gcol = gcol;
    j.pose = pose;
    for (int i = 0; i < 3; i++)
        j.pos[i] = sensor_pos_f[i];
    q.seg_tail.store(t + 1, std::memory_order_release);
}

void Oracle::pipe_push_associate(int64_t gcol)
{
    Pipe& q = *pipe;
    const uint64_t t = q.assoc_tail.load(std::memory_order_relaxed);
    while (t - q.assoc_head.load(std::memory_order_acquire) >= Pipe::CAP && !q.failed.load())
        std::this_thread::yield();
    q.assoc[t % Pipe::CAP] = gcol;
    q.assoc_tail.store(t + 1, std::memory_order_release);
}

// runs n firings through the three-thread pipeline; returns seconds (-1: an exception, its text in o.error)
static double run_pipeline(Oracle& o, int64_t n, const float* xyz, const uint8_t* intensity, const double* poses)
{
    Oracle::Pipe q;
    o.pipe = &q;
    auto fail = [&](const char* what)
    {
        if (!q.failed.exchange(true))
            q.error = what;
    };
    std::thread seg_thread(
        [&]()
        {
            try
            {
                while (true)
                {
                    const uint64_t h = q.seg_head.load(std::memory_order_relaxed);
                    if (h == q.seg_tail.load(std::memory_order_acquire))
                    {
                        if (q.done_insert.load(std::memory_order_acquire) && h == q.seg_tail.load(std::memory_order_acquire))
                            break;
                        if (q.failed.load())
                            break;
                        std::this_thread::yield();
                        continue;
                    }
                    const Oracle::Pipe::SegJob j = q.seg[h % Oracle::Pipe::CAP];
                    q.seg_head.store(h + 1, std::memory_order_release);
                    for (int i = 0; i < 3; i++)
                        o.seg_pos[i] = j.pos[i];
                    o.segment_column(j.gcol, j.pose);
                }
            }
            catch (const std::exception& e)
            {
                fail(e.what());
            }
            q.done_segment.store(true, std::memory_order_release);
        });
    std::thread assoc_thread(
        [&]()
        {
            try
            {
                while (true)
                {
                    const uint64_t h = q.assoc_head.load(std::memory_order_relaxed);
                    if (h == q.assoc_tail.load(std::memory_order_acquire))
                    {
                        if (q.done_segment.load(std::memory_order_acquire) && h == q.assoc_tail.load(std::memory_order_acquire))
                            break;
                        if (q.failed.load())
                            break;
                        std::this_thread::yield();
                        continue;
                    }
                    const int64_t g = q.assoc[h % Oracle::Pipe::CAP];
                    q.assoc_head.store(h + 1, std::memory_order_release);
                    o.associate_column(g);
                }
            }
            catch (const std::exception& e)
            {
                fail(e.what());
            }
        });
    const auto t0 = std::chrono::steady_clock::now();
    try
    {
        for (int64_t i = 0; i < n && !q.failed.load(); i++)
            o.add_firing(xyz + (size_t) i * o.num_rows * 3, intensity + (size_t) i * o.num_rows, poses + (size_t) i * 12);
    }
    catch (const std::exception& e)
    {
        fail(e.what());
    }
    q.done_insert.store(true, std::memory_order_release);
    seg_thread.join();
    assoc_thread.join();
    const auto t1 = std::chrono::steady_clock::now();
    o.pipe = nullptr;
    if (q.failed.load())
    {
        o.error = q.error;
        return -1.0;
    }
    return std::chrono::duration<double>(t1 - t0).count();
}

} // namespace

extern "C" {

struct orc_handle
{
    Oracle o;
};

orc_handle* orc_create(const cc_config* cfg, int num_rows)
{
    auto* h = new orc_handle();
    h->o.set_config(*cfg);
    h->o.reset(num_rows);
    return h;
}

void orc_destroy(orc_handle* h)
{
    delete h;
}

void orc_record(orc_handle* h, int enable)
{
    h->o.record = enable != 0;
}

// keep only (at least) the last n published column snapshots; 0 = all
void orc_keep_published_tail(orc_handle* h, int64_t n)
{
    h->o.keep_tail = n;
}

int orc_set_config(orc_handle* h, const cc_config* cfg)
{
    h->o.set_config(*cfg);
    return CC_OK;
}

int orc_reset(orc_handle* h, int num_rows)
{
    h->o.reset(num_rows);
    return CC_OK;
}

int orc_set_robot_from_sensor(orc_handle* h, const double* tf12)
{
    h->o.robot_from_sensor = iso_from12(tf12);
    h->o.has_robot_tf = true;
    return CC_OK;
}

// returns CC_OK or the CC_ERR_* that corresponds to the std::runtime_error the reference would throw
int orc_add_firings(orc_handle* h, int64_t n, const float* xyz, const uint8_t* intensity, const double* poses)
{
    Oracle& o = h->o;
    try
    {
        for (int64_t i = 0; i < n; i++)
            o.add_firing(xyz + (size_t) i * o.num_rows * 3, intensity + (size_t) i * o.num_rows, poses + (size_t) i * 12);
    }
    catch (const std::runtime_error& e)
    {
        o.error = e.what();
        if (o.error.find("Transform robot frame") != std::string::npos)
            return CC_ERR_NO_ROBOT_TRANSFORM;
        if (o.error.find("not cleared") != std::string::npos)
            return CC_ERR_RING_OVERRUN;
        return CC_ERR_BOOKKEEPING;
    }
    return CC_OK;
}

// Times n firings with recording off; returns seconds (the cpu_baseline leg of bench.py).
double orc_time_firings(orc_handle* h, int64_t n, const float* xyz, const uint8_t* intensity, const double* poses)
{
    Oracle& o = h->o;
    bool rec = o.record;
    o.record = false;
    auto t0 = std::chrono::steady_clock::now();
    try
    {
        for (int64_t i = 0; i < n; i++)
            o.add_firing(xyz + (size_t) i * o.num_rows * 3, intensity + (size_t) i * o.num_rows, poses + (size_t) i * 12);
    }
    catch (const std::runtime_error& e)
    {
        o.error = e.what();
        o.record = rec;
        return -1.0;
    }
    auto t1 = std::chrono::steady_clock::now();
    o.record = rec;
    return std::chrono::duration<double>(t1 - t0).count();
}

// Mode B of BASELINE.md 3: the same n firings through the three-thread stage pipeline (see Oracle::Pipe); recording off; seconds or -1.
double orc_time_firings_pipeline(orc_handle* h, int64_t n, const float* xyz, const uint8_t* intensity, const double* poses)
{
    Oracle& o = h->o;
    const bool rec = o.record;
    o.record = false;
    const double s = run_pipeline(o, n, xyz, intensity, poses);
    o.record = rec;
    return s;
}

// the pipeline with recording ON (tests: its events and published columns must be the single-threaded ones); CC_OK or an error code
int orc_add_firings_pipeline(orc_handle* h, int64_t n, const float* xyz, const uint8_t* intensity, const double* poses)
{
    return run_pipeline(h->o, n, xyz, intensity, poses) < 0 ? CC_ERR_BOOKKEEPING : CC_OK;
}

// Mode A of BASELINE.md 3: per-call latency of addFiring (one firing = one column for KITTI-shaped streams). out_ns[i] = duration of
// call i in nanoseconds (steady_clock around each call; ~25 ns of clock overhead per sample). Returns the total seconds, -1 on error.
double orc_time_each_firing(orc_handle* h, int64_t n, const float* xyz, const uint8_t* intensity, const double* poses, double* out_ns)
{
    Oracle& o = h->o;
    bool rec = o.record;
    o.record = false;
    double total = 0;
    try
    {
        for (int64_t i = 0; i < n; i++)
        {
            auto t0 = std::chrono::steady_clock::now();
            o.add_firing(xyz + (size_t) i * o.num_rows * 3, intensity + (size_t) i * o.num_rows, poses + (size_t) i * 12);
            auto t1 = std::chrono::steady_clock::now();
            const double ns = std::chrono::duration<double, std::nano>(t1 - t0).count();
            out_ns[i] = ns;
            total += ns;
        }
    }
    catch (const std::runtime_error& e)
    {
        o.error = e.what();
        o.record = rec;
        return -1.0;
    }
    o.record = rec;
    return total * 1e-9;
}

const char* orc_last_error(orc_handle* h)
{
    return h->o.error.c_str();
}

int orc_stream_state(orc_handle* h, cc_stream_state* s)
{
    Oracle& o = h->o;
    memset(s, 0, sizeof(*s));
    s->num_rows = o.num_rows;
    s->num_columns = o.num_columns;
    s->ring_buffer_max_columns = o.ring_cols;
    s->reset_required = o.reset_required;
    s->ring_buffer_start_global_column_index = o.ring_start;
    s->ring_buffer_end_global_column_index = o.ring_end;
    s->first_unfinished_global_column_index = o.first_unfinished;
    s->first_unpublished_global_column_index = o.first_unpublished;
    s->cluster_counter = o.cluster_counter;
    s->firings_consumed = o.firings_consumed;
    s->cells_published = o.cells_published;
    s->clusters_finished = o.clusters_finished;
    s->n_unfinished_trees = (int32_t) o.unfinished.size();
    s->error_a = (int64_t) o.exceed_one_rotation;
    s->error_b = (int64_t) o.max_unfinished;
    return CC_OK;
}

int64_t orc_num_events(orc_handle* h)
{
    return (int64_t) h->o.events.size();
}

int orc_drain_events(orc_handle* h, cc_event* out, int64_t capacity, int64_t* n)
{
    Oracle& o = h->o;
    int64_t k = std::min<int64_t>(capacity, (int64_t) o.events.size());
    for (int64_t i = 0; i < k; i++)
        out[i] = o.events[i];
    o.events.erase(o.events.begin(), o.events.begin() + k);
    *n = k;
    return CC_OK;
}

// Columns as they were at the moment the cluster-view column callback published them.
int orc_read_published(orc_handle* h, int64_t from, int64_t to, const cc_column_view* v)
{
    Oracle& o = h->o;
    if (o.published_base < 0 || from < o.published_base || to >= o.published_base + (int64_t) o.published.size() || to < from)
        return CC_ERR_INVALID_ARGUMENT;
    int R = o.num_rows;
    for (int64_t g = from; g <= to; g++)
    {
        const ColumnSnapshot& s = o.published[(size_t) (g - o.published_base)];
        size_t off = (size_t) (g - from) * R;
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
            if (v->tree_parent_global_column) v->tree_parent_global_column[off + r] = s.parent_gcol[r];
            if (v->tree_parent_row) v->tree_parent_row[off + r] = s.parent_row[r];
        }
    }
    return CC_OK;
}

// Member points of the idx-th cluster that received an id since reset (the order of its CC_EV_CLUSTER event), in the order the reference
// fills cluster_points (cc.cpp:996-1016: trees in BFS order of the tree graph, points of a tree in BFS order of the child lists).
int64_t orc_cluster_members(orc_handle* h, int64_t idx, int64_t capacity, int64_t* gcol, int32_t* row)
{
    Oracle& o = h->o;
    if (idx < 0 || idx >= (int64_t) o.cluster_members.size())
        return -1;
    const auto& m = o.cluster_members[(size_t) idx];
    const int64_t n = std::min<int64_t>(capacity, (int64_t) m.size());
    for (int64_t i = 0; i < n; i++)
    {
        gcol[i] = m[(size_t) i].first;
        row[i] = m[(size_t) i].second;
    }
    return (int64_t) m.size();
}

int64_t orc_published_base(orc_handle* h)
{
    return h->o.published_base;
}

int64_t orc_published_count(orc_handle* h)
{
    return (int64_t) h->o.published.size();
}

} // extern "C"
