// ref_loader_driver.cpp — TEST INFRASTRUCTURE. extern "C" driver around the REFERENCE's own continuous_clustering::KittiLoader (compiled from
// /root/reference/src/evaluation/kitti_loader.cpp where it lies), with the signatures of the korc_* exports of oracle/kitti_oracle.cpp, so that
// tests/test_reference_build.py can diff that restatement against the reference itself. Public API only (kitti_loader.hpp:84-170):
// recoverLaserIndices, undoEgoMotionCorrection, generateRangeImage, interpolate, getStartEndTimestampsVelodyne.
//
// Built ONLY by oracle/build_ref.sh and only against a real Eigen3 (the loader needs <Eigen/Geometry>: Quaterniond::slerp, Isometry3d). No
// stand-in headers, ever: without Eigen3 the recipe stops and oracle/kitti_oracle.cpp stays "parity unpinned".
// Output: oracle/_ref/libloader_ref.so (git-ignored).
#include <continuous_clustering/evaluation/kitti_loader.hpp>

#include <cstdint>
#include <cstring>
#include <vector>

using namespace continuous_clustering;

namespace
{
std::vector<KittiPoint> make_points(int64_t n, const float* pts4)
{
    std::vector<KittiPoint> points(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++)
    {
        points[i].x = pts4[4 * i + 0];
        points[i].y = pts4[4 * i + 1];
        points[i].z = pts4[4 * i + 2];
        points[i].i = pts4[4 * i + 3];
    }
    return points;
}

Eigen::Isometry3d from12(const double* m)
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

void to12(const Eigen::Isometry3d& t, double* m)
{
    for (int r = 0; r < 3; r++)
    {
        for (int c = 0; c < 3; c++)
            m[r * 4 + c] = t.linear()(r, c);
        m[r * 4 + 3] = t.translation()(r);
    }
}

std::vector<StampedPose> make_poses(int64_t n, const uint64_t* stamps, const double* poses)
{
    std::vector<StampedPose> out(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++)
    {
        out[i].stamp = stamps[i];
        out[i].pose = from12(poses + 12 * i);
    }
    return out;
}
} // namespace

extern "C" {

// kitti_loader.cpp:48-99; returns 1 where the reference throws
int kref_recover_laser_indices(int64_t n, const float* pts4, uint8_t* laser_out)
{
    KittiLoader loader;
    std::vector<KittiPoint> points = make_points(n, pts4);
    int rc = 0;
    try
    {
        loader.recoverLaserIndices(points);
    }
    catch (const std::exception&)
    {
        rc = 1;
    }
    for (int64_t i = 0; i < n; i++)
        laser_out[i] = points[i].laser_index;
    return rc;
}

// kitti_loader.cpp:177-210 (points in place)
void kref_undo_ego_motion(int64_t n, float* pts4, uint64_t rotation_start_stamp, uint64_t rotation_end_stamp, const double* mid_pose, int64_t n_poses,
                          const uint64_t* stamps, const double* poses)
{
    KittiLoader loader;
    std::vector<KittiPoint> points = make_points(n, pts4);
    loader.undoEgoMotionCorrection(points, rotation_start_stamp, rotation_end_stamp, from12(mid_pose), make_poses(n_poses, stamps, poses));
    for (int64_t i = 0; i < n; i++)
    {
        pts4[4 * i + 0] = points[i].x;
        pts4[4 * i + 1] = points[i].y;
        pts4[4 * i + 2] = points[i].z;
    }
}

// kitti_loader.cpp:101-175: cell_source[row * 2200 + column] = original_kitti_index of the cell's point, -1 if empty
void kref_generate_range_image(int64_t n, const float* pts4, const uint8_t* laser, int shift_cell_if_already_occupied, int32_t* cell_source)
{
    KittiLoader loader;
    std::vector<KittiPoint> points = make_points(n, pts4);
    for (int64_t i = 0; i < n; i++)
        points[i].laser_index = laser ? laser[i] : 0;
    const std::vector<KittiPoint> image = loader.generateRangeImage(points, shift_cell_if_already_occupied != 0);
    for (size_t k = 0; k < image.size(); k++)
        cell_source[k] = image[k].original_kitti_index;
}

// kitti_loader.cpp:297-328
void kref_interpolate(int64_t n_poses, const uint64_t* stamps, const double* poses, uint64_t stamp, double* out12)
{
    KittiLoader loader;
    to12(loader.interpolate(make_poses(n_poses, stamps, poses), stamp).pose, out12);
}

// kitti_loader.cpp:525-540
void kref_start_end_stamps(int64_t n, const uint64_t* timestamps_middle, uint64_t* timestamps_start, uint64_t* timestamps_end)
{
    std::vector<uint64_t> mid(timestamps_middle, timestamps_middle + n), start, end;
    KittiLoader::getStartEndTimestampsVelodyne(mid, start, end);
    for (int64_t i = 0; i < n && i < static_cast<int64_t>(start.size()); i++)
        timestamps_start[i] = start[i];
    for (int64_t i = 0; i < n && i < static_cast<int64_t>(end.size()); i++)
        timestamps_end[i] = end[i];
}

} // extern "C"
