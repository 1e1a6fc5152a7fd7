// dropin_demo.cpp — drives continuous_clustering::ContinuousClustering (the MI355X build) exactly the way the reference's
// src/tools/kitti_demo.cpp:276-313,386-403 drives the reference class: setConfiguration, reset(rows),
// setTransformRobotFrameFromSensorFrame, two callbacks, then addFiring per firing. Inside the cluster-view column callback it
// reads clustering.range_image_ like kitti_demo.cpp:173-224 does and dumps what it sees, so that the Python test can compare
// it with the CPU oracle.
//
//   dropin_demo <in.bin> <out.bin> [batch]
// in.bin : int32 rows, cols, n, kitti(1)/default(0)/default + ego box(2); then n*rows*3 float xyz, n*rows uint8 intensity, n*12 double poses
// out.bin: records, see the writer below
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../continuous_clustering_amd/csrc/continuous_clustering.hpp"

using namespace continuous_clustering;

int main(int argc, char** argv)
{
    if (argc < 3)
        return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f)
        return 2;
    int hdr[4];
    if (fread(hdr, 4, 4, f) != 4)
        return 2;
    const int rows = hdr[0], cols = hdr[1], n = hdr[2], kitti = hdr[3];
    std::vector<float> xyz((size_t) n * rows * 3);
    std::vector<uint8_t> inten((size_t) n * rows);
    std::vector<double> poses((size_t) n * 12);
    if (fread(xyz.data(), 4, xyz.size(), f) != xyz.size() || fread(inten.data(), 1, inten.size(), f) != inten.size() ||
        fread(poses.data(), 8, poses.size(), f) != poses.size())
        return 2;
    fclose(f);
    const int batch = argc > 3 ? atoi(argv[3]) : 1;      // 0 = adaptive batching (setAdaptiveBatching); -1 = the reference's API only, with the
                                                         // reference's default is_single_threaded = false (the asynchronous mode)
    const double rate_hz = argc > 4 ? atof(argv[4]) : 0.;  // > 0: feed the firings paced like a live sensor (firings per second)

    ContinuousClustering clustering;
    Configuration config; // kitti_demo.cpp:279-294
    const bool reference_api_only = batch < 0;
    if (kitti == 1)
    {
        config.general.is_single_threaded = !reference_api_only;
        config.clustering.ignore_points_in_chessboard_pattern = false;
        config.clustering.max_distance = 0.5;
        config.ground_segmentation.height_ref_to_maximum_ = 0.5;
        config.ground_segmentation.height_ref_to_ground_ = -1.7;
        config.ground_segmentation.length_ref_to_front_end_ = 3;
        config.ground_segmentation.length_ref_to_rear_end_ = -3;
        config.ground_segmentation.width_ref_to_left_mirror_ = 1.5;
        config.ground_segmentation.width_ref_to_right_mirror_ = -1.5;
    }
    else if (kitti == 2) // library defaults + an ego box (tests/cases.py `_vls`)
    {
        config.general.is_single_threaded = !reference_api_only;
        config.ground_segmentation.height_ref_to_maximum_ = 0.5;
        config.ground_segmentation.height_ref_to_ground_ = -1.7;
        config.ground_segmentation.length_ref_to_front_end_ = 3;
        config.ground_segmentation.length_ref_to_rear_end_ = -3;
        config.ground_segmentation.width_ref_to_left_mirror_ = 1.5;
        config.ground_segmentation.width_ref_to_right_mirror_ = -1.5;
    }
    config.range_image.num_columns = cols;
    clustering.setConfiguration(config);
    clustering.reset(rows);
    clustering.setTransformRobotFrameFromSensorFrame(Pose3d::Identity());
    if (batch == 0)
        clustering.setAdaptiveBatching();
    else if (batch > 0)
        clustering.setBatchSize(batch);
    // (batch < 0: nothing but the reference's own calls)

    using clk = std::chrono::steady_clock;
    std::vector<double> due_us((size_t) n, 0.), seen_us((size_t) n, -1.);
    const auto t0 = clk::now();
    auto now_us = [&]() { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); };
    const bool dump = std::string(argv[2]) != "/dev/null"; // timing runs: callbacks installed, nothing written
    FILE* out = fopen(argv[2], "wb");
    long long n_ground_cb = 0, n_cluster_cb = 0, n_ground_view_violations = 0, n_published = 0;
    clustering.setFinishedColumnCallback(
        [&](int64_t from, int64_t to, bool ground_points_only)
        {
            if (ground_points_only)
            {
                // asynchronous mode: the callbacks run on the class's worker thread. The j-th ground-view callback is caused by firing j + 1
                // (one column per firing for the streams this harness feeds): that firing has been delivered now
                if (reference_api_only && n_ground_cb + 1 < n && seen_us[(size_t) n_ground_cb + 1] < 0)
                    seen_us[(size_t) n_ground_cb + 1] = now_us();
                n_ground_cb++;
                // the ground view runs before the column is associated (cc.cpp:618-623): clustering fields still as clearColumns left them
                int lc = static_cast<int>(from % clustering.ring_buffer_max_columns);
                for (int r = 0; r < clustering.num_rows_; r++)
                {
                    const Point& p = clustering.range_image_[lc * clustering.num_rows_ + r];
                    if (p.global_column_index != from || p.tree_root_.column_index != -1 || p.id != 0 || !p.child_points.empty() ||
                        p.number_of_visited_neighbors != 0 || p.ground_point_label == 0)
                        n_ground_view_violations++;
                }
                return;
            }
            if (!dump)
            {
                n_published += to >= from ? to - from + 1 : 0;
                return;
            }
            // record type 1: published range, then per cell what kitti_demo reads (kitti_demo.cpp:192-216) + a few more fields
            int32_t tag = 1;
            fwrite(&tag, 4, 1, out);
            fwrite(&from, 8, 1, out);
            fwrite(&to, 8, 1, out);
            for (int64_t g = from; g <= to; g++)
            {
                int lc = static_cast<int>(g % clustering.ring_buffer_max_columns);
                for (int r = 0; r < clustering.num_rows_; r++)
                {
                    const Point& p = clustering.range_image_[lc * clustering.num_rows_ + r];
                    uint64_t id = p.id, uidx = p.globally_unique_point_index, stamp = p.stamp;
                    uint8_t lab[3] = {p.ground_point_label, p.debug_ground_point_label, (uint8_t) p.is_ignored};
                    float geo[3] = {p.distance, p.inclination_angle, p.azimuth_angle};
                    fwrite(&id, 8, 1, out);
                    fwrite(&uidx, 8, 1, out);
                    fwrite(&stamp, 8, 1, out);
                    fwrite(lab, 1, 3, out);
                    fwrite(geo, 4, 3, out);
                    // what the ROS packers read of the clustering stage (ros_utils.cpp:289-295)
                    double fin = p.finished_at_continuous_azimuth_angle;
                    int32_t more[6] = {(int32_t) p.child_points.size(), (int32_t) p.tree_root_.row_index, (int32_t) p.tree_root_.column_index,
                                       p.number_of_visited_neighbors, (int32_t) p.belongs_to_finished_cluster, (int32_t) p.tree_num_points};
                    uint64_t tree_id = p.tree_id;
                    fwrite(&fin, 8, 1, out);
                    fwrite(more, 4, 6, out);
                    fwrite(&tree_id, 8, 1, out);
                }
            }
        });
    clustering.setFinishedClusterCallback(
        [&](const std::vector<Point>& pts, uint64_t stamp)
        {
            n_cluster_cb++;
            if (!dump)
                return;
            int32_t tag = 2;
            uint64_t cnt = pts.size(), id = pts.empty() ? 0 : pts[0].id;
            fwrite(&tag, 4, 1, out);
            fwrite(&id, 8, 1, out);
            fwrite(&cnt, 8, 1, out);
            fwrite(&stamp, 8, 1, out);
            // the member points in the order of the vector (cc.cpp:996-1016), and whether every copy carries the cluster id (cc.cpp:1005)
            for (const Point& p : pts)
            {
                int64_t g = p.global_column_index;
                int32_t r = p.row_index, ok = p.id == id ? 1 : 0;
                fwrite(&g, 8, 1, out);
                fwrite(&r, 4, 1, out);
                fwrite(&ok, 4, 1, out);
            }
        });

    // real-time feed: firing k is due at t0 + k / rate; its latency is measured from that instant to the moment its ground-view callback
    // has run (the column it finishes has been segmented and handed on) — 0 pacing = as fast as the class takes them
    long long delivered_upto = 0; // firings [0, delivered_upto) have been through the engine
    for (int k = 0; k < n; k++)
    {
        if (rate_hz > 0)
        {
            due_us[(size_t) k] = 1e6 * k / rate_hz;
            while (now_us() < due_us[(size_t) k])
            {
            }
        }
        else
            due_us[(size_t) k] = now_us();
        RawPoints::Ptr firing(new RawPoints);
        firing->stamp = 1000000ull + 45ull * k;
        firing->points.resize(rows);
        for (int r = 0; r < rows; r++)
        {
            RawPoint& q = firing->points[r];
            q.x = xyz[((size_t) k * rows + r) * 3 + 0];
            q.y = xyz[((size_t) k * rows + r) * 3 + 1];
            q.z = xyz[((size_t) k * rows + r) * 3 + 2];
            q.intensity = inten[(size_t) k * rows + r];
            q.stamp = firing->stamp;
            q.firing_index = k;
            q.globally_unique_point_index = ((uint64_t) k << 16) | (uint64_t) r; // kitti_demo.cpp:153-155 style tag
        }
        Pose3d pose;
        for (int i = 0; i < 12; i++)
            pose.m[i] = poses[(size_t) k * 12 + i];
        const long long before = n_ground_cb;
        clustering.addFiring(firing, pose);
        if (!reference_api_only && (n_ground_cb != before || batch == 1))
        {
            // the call went through the engine: every firing fed so far has been delivered
            const double t = now_us();
            for (long long j = delivered_upto; j <= k; j++)
                seen_us[(size_t) j] = t;
            delivered_upto = k + 1;
        }
    }
    clustering.flush();
    {
        const double t = now_us();
        for (long long j = delivered_upto; j < n; j++)
            if (seen_us[(size_t) j] < 0)
                seen_us[(size_t) j] = t;
        std::vector<double> lat;
        for (int k = n / 10; k < n; k++) // skip the start-up (ring not started, graph capture)
            lat.push_back(seen_us[(size_t) k] - due_us[(size_t) k]);
        std::sort(lat.begin(), lat.end());
        const double total_s = t * 1e-6;
        // stalls: runs of consecutive firings delivered more than 2 ms late (one hiccup of the feeder, the worker or the GPU delays everything queued behind it)
        int stalls = 0;
        for (int k = n / 10; k < n; k++)
        {
            const bool late = seen_us[(size_t) k] - due_us[(size_t) k] > 2000.;
            const bool prev_late = k > n / 10 && seen_us[(size_t) k - 1] - due_us[(size_t) k - 1] > 2000.;
            stalls += late && !prev_late ? 1 : 0;
        }
        printf("feed rate_hz=%.0f batch=%d firings=%d seconds=%.4f firings_per_s=%.0f latency_us_p50=%.1f p99=%.1f max=%.1f p999=%.1f stalls_over_2ms=%d\n", rate_hz, batch, n,
               total_s, n / total_s, lat[lat.size() / 2], lat[(size_t) (lat.size() * 0.99)], lat.back(), lat[(size_t) (lat.size() * 0.999)], stalls);
    }
    int32_t tag = 3;
    fwrite(&tag, 4, 1, out);
    fwrite(&n_ground_cb, 8, 1, out);
    fwrite(&n_cluster_cb, 8, 1, out);
    fwrite(&n_ground_view_violations, 8, 1, out);
    fclose(out);
    // error behaviour: wrong firing size must throw like continuous_clustering.cpp:90-91
    try
    {
        RawPoints::Ptr bad(new RawPoints);
        bad->points.resize(rows + 1);
        clustering.addFiring(bad, Pose3d::Identity());
        return 3;
    }
    catch (const std::runtime_error&)
    {
    }
    printf("ok ground_cb=%lld cluster_cb=%lld\n", n_ground_cb, n_cluster_cb);
    return 0;
}
