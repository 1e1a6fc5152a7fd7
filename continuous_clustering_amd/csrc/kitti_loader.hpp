// kitti_loader.hpp — the reference's KittiLoader API (include/continuous_clustering/evaluation/kitti_loader.hpp:17-170) on top of
// the MI355X C-ABI of include/cc_kitti.h.
//
// Same namespace, struct and method names and argument meaning as the reference, so that src/tools/kitti_demo.cpp:229-395 reads the
// same against this header. File parsing is host code (it is I/O); the per-point methods — recoverLaserIndices,
// undoEgoMotionCorrection, generateRangeImage — run as HIP kernels (cc_kitti_convert_frames) and have no CPU variant; the pose
// arithmetic (interpolate, the products in getAllDynamicTransforms) is the plain-C host part of the same library.
//
// Poses: Eigen::Isometry3d / Affine3d of the reference are the row-major 3x4 Pose3d of continuous_clustering.hpp here.
// The raw-KITTI (oxts) helpers of the reference (kitti_loader.cpp:212-281) are not part of the replay path and are not mirrored.
#pragma once

#include <cmath>
#include <cstdint>
#include <filesystem>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cc_kitti.h"
#include "continuous_clustering.hpp"

using Path = std::filesystem::path;

namespace continuous_clustering
{

struct RawSequenceSubset // kitti_loader.hpp:18-24
{
    std::string day_string;
    std::string sequence_string;
    int first_frame;
    int last_frame;
};

struct KittiPoint // kitti_loader.hpp:27-45
{
    float x{std::nanf("")};
    float y{std::nanf("")};
    float z{std::nanf("")};
    float i{std::nanf("")};
    uint16_t semantic_label{0};
    uint16_t instance_label{0};
    uint8_t laser_index{0};
    int32_t original_kitti_index{-1};
};

struct StampedPose // kitti_loader.hpp:75-79
{
    uint64_t stamp;
    Pose3d pose;
};

class KittiLoader
{
  public:
    static const int NUM_LASERS = 64;
    static const int RANGE_IMAGE_HEIGHT = NUM_LASERS;
    static const int RANGE_IMAGE_WIDTH = 2200;

  public:
    explicit KittiLoader(int hip_device = 0);
    ~KittiLoader();
    KittiLoader(const KittiLoader&) = delete;
    KittiLoader& operator=(const KittiLoader&) = delete;

    // LiDAR
    std::vector<KittiPoint> loadPointCloud(const Path& path);
    void loadSemanticKittiLabels(const Path& path, std::vector<KittiPoint>& points);
    template<typename T>
    static std::vector<T> loadFlattenedPointCloud(const Path& path)
    {
        std::ifstream fs{path, std::ios::in | std::ios::binary | std::ios::ate};
        if (!fs.is_open())
            throw std::runtime_error("Unable to open file: " + path.string());
        const int64_t number_of_bytes{fs.tellg()};
        if (number_of_bytes == -1 || number_of_bytes % sizeof(T) != 0)
            throw std::runtime_error("File seems to be corrupt: " + path.string());
        std::vector<T> flattened(static_cast<size_t>(number_of_bytes) / sizeof(T));
        fs.seekg(0, std::ios::beg);
        fs.read(reinterpret_cast<char*>(flattened.data()), number_of_bytes);
        return flattened;
    }
    void recoverLaserIndices(std::vector<KittiPoint>& points);
    std::vector<KittiPoint> generateRangeImage(const std::vector<KittiPoint>& unorganized_points, bool shift_cell_if_already_occupied = true);
    void undoEgoMotionCorrection(std::vector<KittiPoint>& corrected_points, uint64_t rotation_start_stamp, uint64_t rotation_end_stamp,
                                 const Pose3d& odom_from_velodyne_at_middle_of_rotation, const std::vector<StampedPose>& odom_from_velodyne);

    // Poses
    StampedPose interpolate(const std::vector<StampedPose>& transforms, uint64_t stamp);
    std::vector<StampedPose> getAllDynamicTransforms(const Path& path_poses_file, const std::vector<uint64_t>& timestamps = {},
                                                     const Pose3d& tf_cam0_from_x = Pose3d::Identity());
    void getStaticTransformAndProjectionMatrices(const Path& path_calib_file, Pose3d& tf_cam0_from_velodyne, Pose3d& projection_matrix_cam0,
                                                 Pose3d& projection_matrix_cam1, Pose3d& projection_matrix_cam2, Pose3d& projection_matrix_cam3);

    // Timing
    static std::vector<uint64_t> loadTimestamps(const Path& timestamp_path, bool make_fake_absolute);
    static void getStartEndTimestampsVelodyne(const std::vector<uint64_t>& timestamps_middle, std::vector<uint64_t>& timestamps_start,
                                              std::vector<uint64_t>& timestamps_end);

    // Meta data
    static std::map<int, RawSequenceSubset> getKittiOdometrySequenceToKittiRawMapping();
    static std::map<uint16_t, std::string> getSemanticKittiLabelNumericToLabelNameMapping();
    static std::map<std::string, uint16_t> getSemanticKittiLabelNameToLabelNumericMapping();

    // Utils
    static std::vector<std::string> split(const std::string& s, char delimiter);
    static std::string padWithZeros(int v, int number_of_digits);

    // ---- extension of the MI355X build: all per-point steps of one frame in one device pass (what kitti_demo.cpp:352-377 does with
    // three calls); returns the range image, `points` receives rows and un-corrected coordinates like the three calls would leave them.
    std::vector<KittiPoint> frameToRangeImage(std::vector<KittiPoint>& points, uint64_t rotation_start_stamp, uint64_t rotation_end_stamp,
                                              const Pose3d& odom_from_velodyne_at_middle_of_rotation,
                                              const std::vector<StampedPose>& odom_from_velodyne);

  private:
    void ensure(size_t n_points);
    void check(int rc) const;
    void pack(const std::vector<KittiPoint>& points);
    std::vector<double> binTable(uint64_t start, uint64_t end, const Pose3d& mid, const std::vector<StampedPose>& poses) const;
    std::vector<KittiPoint> organize(const std::vector<KittiPoint>& points) const;
    void reportRows(const cc_kitti_frame_info& info) const;

    int device_{0};
    cc_kitti* handle_{nullptr};
    size_t capacity_{0};
    std::vector<float> xyzi_;
    std::vector<uint8_t> rows_;
    std::vector<int32_t> cells_;
};

} // namespace continuous_clustering
