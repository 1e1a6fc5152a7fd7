// kitti_loader.cpp — host side of the KittiLoader mirror (see kitti_loader.hpp). File formats and error texts follow
// src/evaluation/kitti_loader.cpp of the reference; the per-point work is delegated to the HIP kernels behind include/cc_kitti.h.
#include "kitti_loader.hpp"

#include <chrono>
#include <iomanip>
#include <iostream>
#include <sstream>

namespace continuous_clustering
{

namespace
{
constexpr size_t kCells = static_cast<size_t>(CC_KITTI_ROWS) * CC_KITTI_COLS;

// "name: v1 v2 ... v12" (calib.txt) or "v0 v1 ... v11" (poses.txt) -> row-major 3x4
Pose3d parse3x4(const std::vector<std::string>& v, size_t first)
{
    if (v.size() < first + 12)
        throw std::runtime_error("Expected 12 numbers in line");
    Pose3d p;
    for (int k = 0; k < 12; k++)
        p.m[k] = std::stod(v[first + static_cast<size_t>(k)]);
    return p;
}

void flatten(const std::vector<StampedPose>& poses, std::vector<uint64_t>& stamps, std::vector<double>& m)
{
    stamps.resize(poses.size());
    m.resize(poses.size() * 12);
    for (size_t i = 0; i < poses.size(); i++)
    {
        stamps[i] = poses[i].stamp;
        std::copy(poses[i].pose.m, poses[i].pose.m + 12, m.begin() + static_cast<std::ptrdiff_t>(i * 12));
    }
}
} // namespace

KittiLoader::KittiLoader(int hip_device) : device_(hip_device)
{
}

KittiLoader::~KittiLoader()
{
    if (handle_)
        cc_kitti_destroy(handle_);
}

void KittiLoader::check(int rc) const
{
    if (rc != CC_OK)
        throw std::runtime_error(std::string("cc_kitti: ") + cc_kitti_last_error());
}

void KittiLoader::ensure(size_t n_points)
{
    if (handle_ && n_points <= capacity_)
        return;
    if (handle_)
        cc_kitti_destroy(handle_);
    handle_ = nullptr;
    capacity_ = std::max<size_t>(n_points + n_points / 4, 160000);
    check(cc_kitti_create(&handle_, device_, 1, static_cast<int64_t>(capacity_), nullptr));
}

// ---- files ------------------------------------------------------------------------------------------------------------------

std::vector<KittiPoint> KittiLoader::loadPointCloud(const Path& path)
{
    // .bin: x y z i as float32 per point (kitti_loader.cpp:12-29)
    const std::vector<float> flat = loadFlattenedPointCloud<float>(path);
    std::vector<KittiPoint> points(flat.size() / 4);
    for (size_t k = 0; k < points.size(); k++)
    {
        points[k].x = flat[4 * k];
        points[k].y = flat[4 * k + 1];
        points[k].z = flat[4 * k + 2];
        points[k].i = flat[4 * k + 3];
    }
    return points;
}

void KittiLoader::loadSemanticKittiLabels(const Path& path, std::vector<KittiPoint>& points)
{
    // .label: semantic u16, instance u16 per point (kitti_loader.cpp:31-46)
    const std::vector<uint16_t> flat = loadFlattenedPointCloud<uint16_t>(path);
    const size_t num_points = flat.size() / 2;
    if (num_points != points.size())
        throw std::runtime_error("Number of points does not match (label/bin): " + std::to_string(num_points) + " / " +
                                 std::to_string(points.size()));
    for (size_t k = 0; k < num_points; k++)
    {
        points[k].semantic_label = flat[2 * k];
        points[k].instance_label = flat[2 * k + 1];
    }
}

std::vector<uint64_t> KittiLoader::loadTimestamps(const Path& timestamp_path, bool make_fake_absolute)
{
    // times.txt: seconds since sequence start, one per line (kitti_loader.cpp:498-523)
    uint64_t fake_start_stamp = 0;
    if (make_fake_absolute)
        fake_start_stamp = static_cast<uint64_t>(
            std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count());
    std::ifstream is(timestamp_path);
    if (!is.is_open())
        throw std::runtime_error("File does not exist: " + timestamp_path.string());
    std::vector<uint64_t> timestamps;
    std::string line;
    while (std::getline(is, line))
    {
        const double dt = std::stod(line);
        timestamps.push_back(fake_start_stamp + static_cast<uint64_t>(dt * 1000000000UL));
    }
    return timestamps;
}

void KittiLoader::getStartEndTimestampsVelodyne(const std::vector<uint64_t>& timestamps_middle, std::vector<uint64_t>& timestamps_start,
                                                std::vector<uint64_t>& timestamps_end)
{
    timestamps_start.assign(timestamps_middle.size(), 0);
    timestamps_end.assign(timestamps_middle.size(), 0);
    if (timestamps_middle.empty())
        return;
    if (cc_kitti_start_end_stamps(static_cast<int64_t>(timestamps_middle.size()), timestamps_middle.data(), timestamps_start.data(),
                                  timestamps_end.data()) != CC_OK)
        throw std::runtime_error(std::string("cc_kitti: ") + cc_kitti_last_error());
}

std::vector<StampedPose> KittiLoader::getAllDynamicTransforms(const Path& path_poses_file, const std::vector<uint64_t>& timestamps,
                                                              const Pose3d& tf_cam0_from_x)
{
    // poses.txt: first_cam0_from_cam0 as 12 numbers per line (kitti_loader.cpp:330-369)
    std::ifstream is(path_poses_file);
    if (!is.is_open())
        throw std::runtime_error("Unable to open poses file: " + path_poses_file.string());
    std::vector<StampedPose> poses;
    std::string line;
    size_t i = 0;
    while (std::getline(is, line) && (timestamps.empty() || i < timestamps.size()))
    {
        const Pose3d row = parse3x4(split(line, ' '), 0);
        StampedPose sp{timestamps.empty() ? 0 : timestamps[i], Pose3d()};
        check(cc_kitti_pose_from_line(row.m, tf_cam0_from_x.m, sp.pose.m));
        poses.push_back(sp);
        i++;
    }
    if (!timestamps.empty() && i != timestamps.size())
        throw std::runtime_error("The number of poses (i.e. lines in poses.txt) does not match with number of timestamps.");
    return poses;
}

void KittiLoader::getStaticTransformAndProjectionMatrices(const Path& path_calib_file, Pose3d& tf_cam0_from_velodyne, Pose3d& projection_matrix_cam0,
                                                          Pose3d& projection_matrix_cam1, Pose3d& projection_matrix_cam2,
                                                          Pose3d& projection_matrix_cam3)
{
    // calib.txt: P0..P3 and Tr, "name: 12 numbers" each (kitti_loader.cpp:371-425)
    std::ifstream is(path_calib_file);
    if (!is.is_open())
        throw std::runtime_error("Unable to open calibration file: " + path_calib_file.string());
    Pose3d* targets[5] = {&projection_matrix_cam0, &projection_matrix_cam1, &projection_matrix_cam2, &projection_matrix_cam3,
                          &tf_cam0_from_velodyne};
    std::string line;
    for (Pose3d* t : targets)
    {
        std::getline(is, line);
        *t = parse3x4(split(line, ' '), 1);
    }
}

// ---- poses ------------------------------------------------------------------------------------------------------------------

StampedPose KittiLoader::interpolate(const std::vector<StampedPose>& transforms, uint64_t stamp)
{
    std::vector<uint64_t> stamps;
    std::vector<double> m;
    flatten(transforms, stamps, m);
    StampedPose out{stamp, Pose3d()};
    check(cc_kitti_pose_interpolate(static_cast<int64_t>(stamps.size()), stamps.data(), m.data(), stamp, out.pose.m));
    return out;
}

std::vector<double> KittiLoader::binTable(uint64_t start, uint64_t end, const Pose3d& mid, const std::vector<StampedPose>& poses) const
{
    std::vector<uint64_t> stamps;
    std::vector<double> m;
    flatten(poses, stamps, m);
    std::vector<double> table(512 * 12);
    int32_t num_bins = 0;
    check(cc_kitti_bin_transforms(static_cast<int64_t>(stamps.size()), stamps.data(), m.data(), start, end, mid.m, table.data(), 512, &num_bins));
    table.resize(static_cast<size_t>(num_bins) * 12);
    return table;
}

// ---- per-point steps on the GPU ---------------------------------------------------------------------------------------------

void KittiLoader::pack(const std::vector<KittiPoint>& points)
{
    xyzi_.resize(points.size() * 4);
    rows_.resize(points.size());
    for (size_t k = 0; k < points.size(); k++)
    {
        xyzi_[4 * k] = points[k].x;
        xyzi_[4 * k + 1] = points[k].y;
        xyzi_[4 * k + 2] = points[k].z;
        xyzi_[4 * k + 3] = points[k].i;
        rows_[k] = points[k].laser_index;
    }
}

void KittiLoader::reportRows(const cc_kitti_frame_info& info) const
{
    if (info.rows_found != RANGE_IMAGE_HEIGHT)
        std::cerr << "Wrong number of rows found: " << std::to_string(info.rows_found) << std::endl; // kitti_loader.cpp:92-94
    if (info.max_columns > RANGE_IMAGE_WIDTH)
        throw std::runtime_error("More points in a single row than expected: " + std::to_string(info.max_columns)); // :96-97
}

void KittiLoader::recoverLaserIndices(std::vector<KittiPoint>& points)
{
    ensure(points.size());
    pack(points);
    cc_kitti_frame f{};
    f.points = xyzi_.data();
    f.n_points = static_cast<int64_t>(points.size());
    f.stages = CC_KITTI_RECOVER_ROWS;
    check(cc_kitti_convert_frames(handle_, 1, &f));
    cc_kitti_frame_info info{};
    check(cc_kitti_frame_result(handle_, 0, &info, nullptr, rows_.data(), nullptr));
    // the reference leaves laser_index untouched from the break on (kitti_loader.cpp:74-76); a fresh cloud has 0 there
    for (size_t k = 0; k < points.size() && static_cast<int64_t>(k) < info.break_index; k++)
        points[k].laser_index = rows_[k];
    reportRows(info);
}

void KittiLoader::undoEgoMotionCorrection(std::vector<KittiPoint>& corrected_points, uint64_t rotation_start_stamp, uint64_t rotation_end_stamp,
                                          const Pose3d& odom_from_velodyne_at_middle_of_rotation,
                                          const std::vector<StampedPose>& odom_from_velodyne)
{
    const std::vector<double> table = binTable(rotation_start_stamp, rotation_end_stamp, odom_from_velodyne_at_middle_of_rotation, odom_from_velodyne);
    if (table.empty())
        return;
    ensure(corrected_points.size());
    pack(corrected_points);
    cc_kitti_frame f{};
    f.points = xyzi_.data();
    f.n_points = static_cast<int64_t>(corrected_points.size());
    f.stages = CC_KITTI_UNDO_EGO_MOTION;
    f.rotation_start_stamp = rotation_start_stamp;
    f.rotation_end_stamp = rotation_end_stamp;
    f.bin_transforms = table.data();
    f.num_bins = static_cast<int32_t>(table.size() / 12);
    check(cc_kitti_convert_frames(handle_, 1, &f));
    check(cc_kitti_frame_result(handle_, 0, nullptr, xyzi_.data(), nullptr, nullptr));
    for (size_t k = 0; k < corrected_points.size(); k++)
    {
        corrected_points[k].x = xyzi_[4 * k];
        corrected_points[k].y = xyzi_[4 * k + 1];
        corrected_points[k].z = xyzi_[4 * k + 2];
    }
}

std::vector<KittiPoint> KittiLoader::organize(const std::vector<KittiPoint>& points) const
{
    std::vector<KittiPoint> organized(kCells, KittiPoint());
    for (size_t cell = 0; cell < kCells; cell++)
    {
        const int32_t src = cells_[cell];
        if (src >= 0)
        {
            organized[cell] = points[static_cast<size_t>(src)];
            organized[cell].original_kitti_index = src;
        }
    }
    return organized;
}

std::vector<KittiPoint> KittiLoader::generateRangeImage(const std::vector<KittiPoint>& unorganized_points, bool shift_cell_if_already_occupied)
{
    ensure(unorganized_points.size());
    pack(unorganized_points);
    cc_kitti_frame f{};
    f.points = xyzi_.data();
    f.n_points = static_cast<int64_t>(unorganized_points.size());
    f.laser_index = rows_.data();
    f.stages = CC_KITTI_RANGE_IMAGE | (shift_cell_if_already_occupied ? CC_KITTI_SHIFT_OCCUPIED : 0u);
    check(cc_kitti_convert_frames(handle_, 1, &f));
    cells_.resize(kCells);
    check(cc_kitti_frame_result(handle_, 0, nullptr, nullptr, nullptr, cells_.data()));
    return organize(unorganized_points);
}

std::vector<KittiPoint> KittiLoader::frameToRangeImage(std::vector<KittiPoint>& points, uint64_t rotation_start_stamp, uint64_t rotation_end_stamp,
                                                       const Pose3d& odom_from_velodyne_at_middle_of_rotation,
                                                       const std::vector<StampedPose>& odom_from_velodyne)
{
    const std::vector<double> table = binTable(rotation_start_stamp, rotation_end_stamp, odom_from_velodyne_at_middle_of_rotation, odom_from_velodyne);
    ensure(points.size());
    pack(points);
    cc_kitti_frame f{};
    f.points = xyzi_.data();
    f.n_points = static_cast<int64_t>(points.size());
    f.stages = CC_KITTI_RECOVER_ROWS | CC_KITTI_RANGE_IMAGE | CC_KITTI_SHIFT_OCCUPIED | (table.empty() ? 0u : CC_KITTI_UNDO_EGO_MOTION);
    f.rotation_start_stamp = rotation_start_stamp;
    f.rotation_end_stamp = rotation_end_stamp;
    f.bin_transforms = table.empty() ? nullptr : table.data();
    f.num_bins = static_cast<int32_t>(table.size() / 12);
    check(cc_kitti_convert_frames(handle_, 1, &f));
    cells_.resize(kCells);
    cc_kitti_frame_info info{};
    check(cc_kitti_frame_result(handle_, 0, &info, xyzi_.data(), rows_.data(), cells_.data()));
    for (size_t k = 0; k < points.size(); k++)
    {
        points[k].x = xyzi_[4 * k];
        points[k].y = xyzi_[4 * k + 1];
        points[k].z = xyzi_[4 * k + 2];
        points[k].laser_index = rows_[k];
    }
    reportRows(info);
    return organize(points);
}

// ---- meta data (kitti_loader.cpp:542-612) -------------------------------------------------------------------------------------

std::map<int, RawSequenceSubset> KittiLoader::getKittiOdometrySequenceToKittiRawMapping()
{
    // KITTI odometry sequence -> raw drive and frame range (devkit_raw_data readme)
    static const struct
    {
        int seq;
        const char* day;
        const char* drive;
        int first, last;
    } table[] = {{0, "2011_10_03", "0027", 0, 4540}, {1, "2011_10_03", "0042", 0, 1100},   {2, "2011_10_03", "0034", 0, 4660},
                 {3, "2011_09_26", "0067", 0, 800},  {4, "2011_09_30", "0016", 0, 270},    {5, "2011_09_30", "0018", 0, 2760},
                 {6, "2011_09_30", "0020", 0, 1100}, {7, "2011_09_30", "0027", 0, 1100},   {8, "2011_09_30", "0028", 1100, 5170},
                 {9, "2011_09_30", "0033", 0, 1590}, {10, "2011_09_30", "0034", 0, 1200}};
    std::map<int, RawSequenceSubset> map;
    for (const auto& t : table)
        map.insert({t.seq, {t.day, std::string(t.day) + "_drive_" + t.drive + "_sync", t.first, t.last}});
    return map;
}

std::map<uint16_t, std::string> KittiLoader::getSemanticKittiLabelNumericToLabelNameMapping()
{
    // semantic-kitti.yaml label ids
    static const std::pair<uint16_t, const char*> names[] = {
        {0, "unlabeled"},      {1, "outlier"},       {10, "car"},           {11, "bicycle"},          {13, "bus"},
        {15, "motorcycle"},    {16, "on-rails"},     {18, "truck"},         {20, "other-vehicle"},    {30, "person"},
        {31, "bicyclist"},     {32, "motorcyclist"}, {40, "road"},          {44, "parking"},          {48, "sidewalk"},
        {49, "other-ground"},  {50, "building"},     {51, "fence"},         {52, "other-structure"},  {60, "lane-marking"},
        {70, "vegetation"},    {71, "trunk"},        {72, "terrain"},       {80, "pole"},             {81, "traffic-sign"},
        {99, "other-object"},  {252, "moving-car"},  {253, "moving-bicyclist"}, {254, "moving-person"}, {255, "moving-motorcyclist"},
        {256, "moving-on-rails"}, {257, "moving-bus"}, {258, "moving-truck"}, {259, "moving-other-vehicle"}};
    std::map<uint16_t, std::string> map;
    for (const auto& n : names)
        map.insert({n.first, n.second});
    return map;
}

std::map<std::string, uint16_t> KittiLoader::getSemanticKittiLabelNameToLabelNumericMapping()
{
    std::map<std::string, uint16_t> inverse;
    for (const auto& p : getSemanticKittiLabelNumericToLabelNameMapping())
        inverse.insert({p.second, p.first});
    return inverse;
}

std::vector<std::string> KittiLoader::split(const std::string& s, char delimiter)
{
    std::vector<std::string> result;
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, delimiter))
        result.push_back(item);
    return result;
}

std::string KittiLoader::padWithZeros(int v, int number_of_digits)
{
    std::stringstream ss;
    ss << std::setfill('0') << std::setw(number_of_digits) << v;
    return ss.str();
}

} // namespace continuous_clustering
