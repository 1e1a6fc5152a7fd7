// kitti_oracle.cpp — TEST INFRASTRUCTURE (not shipped, not on the product path).
//
// CPU restatement of the KITTI replay path that feeds ContinuousClustering::addFiring in the reference's no-ROS harness:
// KittiLoader (src/evaluation/kitti_loader.cpp) + the pseudo-firing builder of src/tools/kitti_demo.cpp. Single-threaded,
// array-of-structs, same loop order as the reference; every function cites the lines it follows.
//
// ** PARITY UNPINNED ** for everything that goes through Eigen (interpolate, the bin table, pose products): Eigen3 is not in
// this image, so Eigen::Isometry3d / Quaterniond of Eigen 3.4.0 (the version that ships with the reference's Ubuntu 22.04 /
// glibc 2.35 runtime) are restated from its published algorithms (Geometry/Quaternion.h: matrix -> quaternion, slerp,
// toRotationMatrix; Geometry/Transform.h: Isometry product, inverse, rotation() == linear()). The reference's tests hold no
// vectors for this path. The per-point parts only use glibc's atan2f (pinned: oracle/libm_pin.cpp) and IEEE double arithmetic.
//
// Where the reference has undefined behaviour the restatement picks the x86-64/gcc outcome when there is one and says so:
//   * undoEgoMotionCorrection indexes one past the bin table when a point lies exactly at azimuth -pi and the rotation lasts a
//     whole number of milliseconds (:201-204) -> clamped to the last bin;
//   * a point with a NaN azimuth makes generateRangeImage index organized_points with INT_MIN (:121,:165) -> the point is skipped;
//   * static_cast<uint8_t>(NaN or out-of-int32 float) (kitti_demo.cpp:148) -> cvttss2si's 0x80000000, low byte 0.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace
{

constexpr int RANGE_IMAGE_HEIGHT = 64;  // kitti_loader.hpp:85
constexpr int RANGE_IMAGE_WIDTH = 2200; // kitti_loader.hpp:86

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

// ---- the slice of Eigen the path uses -------------------------------------------------------------------------------------
struct Vec3
{
    double v[3];
};
struct Mat3
{
    double a[3][3];
};
struct Isometry
{
    Mat3 linear;
    Vec3 translation;
};

Vec3 mat_vec(const Mat3& m, const Vec3& p)
{
    Vec3 r;
    for (int row = 0; row < 3; row++)
    {
        double acc = m.a[row][0] * p.v[0];
        acc = acc + m.a[row][1] * p.v[1];
        acc = acc + m.a[row][2] * p.v[2];
        r.v[row] = acc;
    }
    return r;
}

Mat3 mat_mat(const Mat3& l, const Mat3& r)
{
    Mat3 out;
    for (int col = 0; col < 3; col++)
    {
        const Vec3 c = mat_vec(l, Vec3{{r.a[0][col], r.a[1][col], r.a[2][col]}});
        for (int row = 0; row < 3; row++)
            out.a[row][col] = c.v[row];
    }
    return out;
}

Isometry compose(const Isometry& l, const Isometry& r)
{
    Isometry out;
    out.linear = mat_mat(l.linear, r.linear);
    const Vec3 lt = mat_vec(l.linear, r.translation);
    for (int k = 0; k < 3; k++)
        out.translation.v[k] = lt.v[k] + l.translation.v[k];
    return out;
}

Isometry inverse(const Isometry& t)
{
    Isometry out;
    Mat3 neg;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
        {
            out.linear.a[r][c] = t.linear.a[c][r];
            neg.a[r][c] = -t.linear.a[c][r];
        }
    out.translation = mat_vec(neg, t.translation);
    return out;
}

Vec3 apply(const Isometry& t, const Vec3& p)
{
    Vec3 r = mat_vec(t.linear, p);
    for (int k = 0; k < 3; k++)
        r.v[k] = r.v[k] + t.translation.v[k];
    return r;
}

struct Quaternion
{
    double x, y, z, w;
    double& at(int k)
    {
        return k == 0 ? x : (k == 1 ? y : z);
    }
};

Quaternion to_quaternion(const Mat3& m)
{
    Quaternion q{};
    double t = m.a[0][0] + m.a[1][1] + m.a[2][2];
    if (t > 0)
    {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m.a[2][1] - m.a[1][2]) * t;
        q.y = (m.a[0][2] - m.a[2][0]) * t;
        q.z = (m.a[1][0] - m.a[0][1]) * t;
        return q;
    }
    int i = 0;
    if (m.a[1][1] > m.a[0][0])
        i = 1;
    if (m.a[2][2] > m.a[i][i])
        i = 2;
    int j = (i + 1) % 3;
    int k = (j + 1) % 3;
    t = std::sqrt(m.a[i][i] - m.a[j][j] - m.a[k][k] + 1.0);
    q.at(i) = 0.5 * t;
    t = 0.5 / t;
    q.w = (m.a[k][j] - m.a[j][k]) * t;
    q.at(j) = (m.a[j][i] + m.a[i][j]) * t;
    q.at(k) = (m.a[k][i] + m.a[i][k]) * t;
    return q;
}

Quaternion slerp(const Quaternion& from, double t, const Quaternion& to)
{
    const double one = 1.0 - std::numeric_limits<double>::epsilon();
    // Vector4d dot product with 2-wide packets: lanes {x,y} + {z,w}, then the horizontal add
    const double lane0 = from.x * to.x + from.z * to.z;
    const double lane1 = from.y * to.y + from.w * to.w;
    const double d = lane0 + lane1;
    const double abs_d = std::abs(d);
    double scale0, scale1;
    if (abs_d >= one)
    {
        scale0 = 1.0 - t;
        scale1 = t;
    }
    else
    {
        const double theta = std::acos(abs_d);
        const double sin_theta = std::sin(theta);
        scale0 = std::sin((1.0 - t) * theta) / sin_theta;
        scale1 = std::sin((t * theta)) / sin_theta;
    }
    if (d < 0)
        scale1 = -scale1;
    return Quaternion{scale0 * from.x + scale1 * to.x, scale0 * from.y + scale1 * to.y, scale0 * from.z + scale1 * to.z,
                      scale0 * from.w + scale1 * to.w};
}

Mat3 to_rotation_matrix(const Quaternion& q)
{
    const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    Mat3 r;
    r.a[0][0] = 1.0 - (tyy + tzz);
    r.a[0][1] = txy - twz;
    r.a[0][2] = txz + twy;
    r.a[1][0] = txy + twz;
    r.a[1][1] = 1.0 - (txx + tzz);
    r.a[1][2] = tyz - twx;
    r.a[2][0] = txz - twy;
    r.a[2][1] = tyz + twx;
    r.a[2][2] = 1.0 - (txx + tyy);
    return r;
}

Isometry from12(const double* m)
{
    Isometry t;
    for (int r = 0; r < 3; r++)
    {
        for (int c = 0; c < 3; c++)
            t.linear.a[r][c] = m[r * 4 + c];
        t.translation.v[r] = m[r * 4 + 3];
    }
    return t;
}

void to12(const Isometry& t, double* m)
{
    for (int r = 0; r < 3; r++)
    {
        for (int c = 0; c < 3; c++)
            m[r * 4 + c] = t.linear.a[r][c];
        m[r * 4 + 3] = t.translation.v[r];
    }
}

struct StampedPose // kitti_loader.hpp:75-79
{
    uint64_t stamp;
    Isometry pose;
};

// KittiLoader::interpolate, kitti_loader.cpp:297-328
Isometry interpolate(const std::vector<StampedPose>& transforms, uint64_t stamp)
{
    auto it_pose_after = std::lower_bound(transforms.begin(), transforms.end(), stamp,
                                          [](const StampedPose& pose, uint64_t s) { return pose.stamp < s; });
    if (it_pose_after == transforms.end())
        return (it_pose_after - 1)->pose;
    if (it_pose_after == transforms.begin())
        return it_pose_after->pose;
    auto it_pose_before = it_pose_after - 1;
    const double f = static_cast<double>(stamp - it_pose_before->stamp) / static_cast<double>(it_pose_after->stamp - it_pose_before->stamp);
    const Quaternion q_before = to_quaternion(it_pose_before->pose.linear);
    const Quaternion q_after = to_quaternion(it_pose_after->pose.linear);
    const Quaternion q = slerp(q_before, f, q_after);
    Isometry out;
    out.linear = to_rotation_matrix(q);
    for (int k = 0; k < 3; k++)
        out.translation.v[k] = (1 - f) * it_pose_before->pose.translation.v[k] + f * it_pose_after->pose.translation.v[k];
    return out;
}

std::vector<StampedPose> make_poses(int64_t n, const uint64_t* stamps, const double* poses)
{
    std::vector<StampedPose> v(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++)
        v[i] = StampedPose{stamps[i], from12(poses + i * 12)};
    return v;
}

std::vector<KittiPoint> make_points(int64_t n, const float* pts4)
{
    // loadPointCloud, kitti_loader.cpp:12-29
    std::vector<KittiPoint> points(static_cast<size_t>(n));
    int64_t flattened_index = 0;
    for (int64_t i = 0; i < n; i++)
    {
        KittiPoint& p = points[i];
        p.x = pts4[flattened_index++];
        p.y = pts4[flattened_index++];
        p.z = pts4[flattened_index++];
        p.i = pts4[flattened_index++];
    }
    return points;
}

} // namespace

extern "C" {

// recoverLaserIndices, kitti_loader.cpp:48-99. Returns 1 where the reference throws ("More points in a single row than expected").
int korc_recover_laser_indices(int64_t n, const float* pts4, uint8_t* laser_out, int32_t* rows_found, int32_t* max_columns_out)
{
    std::vector<KittiPoint> points = make_points(n, pts4);
    int laser_index = 0;
    double prev_azimuth_monotonic = -1;
    int collected_points_for_this_row = 0;
    int max_columns = 0;
    for (auto& point : points)
    {
        double cur_azimuth = std::atan2(point.y, point.x); // float overload
        double cur_azimuth_monotonic = cur_azimuth < 0 ? cur_azimuth + (2 * M_PI) : cur_azimuth;
        if (prev_azimuth_monotonic >= 0 and cur_azimuth_monotonic - prev_azimuth_monotonic < -0.7)
        {
            laser_index++;
            if (laser_index >= RANGE_IMAGE_HEIGHT)
                break;
            max_columns = std::max(max_columns, collected_points_for_this_row);
            collected_points_for_this_row = 0;
        }
        point.laser_index = static_cast<uint8_t>(laser_index);
        prev_azimuth_monotonic = cur_azimuth_monotonic;
        collected_points_for_this_row++;
    }
    for (int64_t i = 0; i < n; i++)
        laser_out[i] = points[i].laser_index;
    *rows_found = laser_index + 1;
    *max_columns_out = max_columns;
    return max_columns > RANGE_IMAGE_WIDTH ? 1 : 0;
}

// the bin table of undoEgoMotionCorrection, kitti_loader.cpp:183-197
int32_t korc_bin_transforms(int64_t n_poses, const uint64_t* stamps, const double* poses, uint64_t rotation_start_stamp,
                            uint64_t rotation_end_stamp, const double* mid_pose, double* out, int32_t capacity)
{
    const std::vector<StampedPose> odom_from_velodyne = make_poses(n_poses, stamps, poses);
    const Isometry odom_from_velodyne_at_middle_of_rotation = from12(mid_pose);
    uint64_t bin_resolution = 1000000;
    uint64_t duration = rotation_end_stamp - rotation_start_stamp;
    int num_bins = static_cast<int>(std::ceil(static_cast<double>(duration) / static_cast<double>(bin_resolution)));
    for (int bin_index = 0; bin_index < num_bins && bin_index < capacity; bin_index++)
    {
        uint64_t stamp_at_bin = rotation_start_stamp + bin_index * bin_resolution + (bin_resolution / 2);
        to12(compose(inverse(interpolate(odom_from_velodyne, stamp_at_bin)), odom_from_velodyne_at_middle_of_rotation), out + bin_index * 12);
    }
    return num_bins;
}

// undoEgoMotionCorrection, kitti_loader.cpp:177-210 (points in place)
void korc_undo_ego_motion(int64_t n, float* pts4, uint64_t rotation_start_stamp, uint64_t rotation_end_stamp, const double* mid_pose,
                          int64_t n_poses, const uint64_t* stamps, const double* poses)
{
    uint64_t bin_resolution = 1000000;
    uint64_t duration = rotation_end_stamp - rotation_start_stamp;
    int num_bins = static_cast<int>(std::ceil(static_cast<double>(duration) / static_cast<double>(bin_resolution)));
    if (num_bins <= 0)
        return;
    std::vector<double> table(static_cast<size_t>(num_bins) * 12);
    korc_bin_transforms(n_poses, stamps, poses, rotation_start_stamp, rotation_end_stamp, mid_pose, table.data(), num_bins);
    for (int64_t k = 0; k < n; k++)
    {
        float* p = pts4 + k * 4;
        double fraction_of_scan_completed = (M_PI - std::atan2(p[1], p[0])) / (2.0 * M_PI);
        double b = (fraction_of_scan_completed * static_cast<double>(duration)) / static_cast<double>(bin_resolution);
        int bin_index = b >= 0 ? (b < num_bins ? static_cast<int>(b) : num_bins - 1) : 0; // header: the clamp replaces UB
        const Isometry tf = from12(table.data() + static_cast<size_t>(bin_index) * 12);
        const Vec3 uncorrected_position = apply(tf, Vec3{{p[0], p[1], p[2]}});
        p[0] = static_cast<float>(uncorrected_position.v[0]);
        p[1] = static_cast<float>(uncorrected_position.v[1]);
        p[2] = static_cast<float>(uncorrected_position.v[2]);
    }
}

// generateRangeImage, kitti_loader.cpp:101-175. cell_source[row * 2200 + column] = original_kitti_index of the cell's point, -1 if
// empty. Returns the number of NaN-azimuth points that were skipped.
int64_t korc_generate_range_image(int64_t n, const float* pts4, const uint8_t* laser, int shift_cell_if_already_occupied, int32_t* cell_source)
{
    std::vector<KittiPoint> unorganized_points = make_points(n, pts4);
    for (int64_t i = 0; i < n; i++)
        unorganized_points[i].laser_index = laser ? laser[i] : 0;
    const double column_width = (2 * M_PI) / RANGE_IMAGE_WIDTH;
    std::vector<KittiPoint> organized_points(RANGE_IMAGE_HEIGHT * RANGE_IMAGE_WIDTH, KittiPoint());
    int64_t skipped = 0;
    for (int32_t original_index = 0; original_index < static_cast<int64_t>(unorganized_points.size()); original_index++)
    {
        const KittiPoint& unorganized_point = unorganized_points[original_index];
        double cur_azimuth = std::atan2(unorganized_point.y, unorganized_point.x);
        if (std::isnan(cur_azimuth))
        {
            skipped++;
            continue;
        }
        auto column_index = static_cast<int>((M_PI - cur_azimuth) / column_width);
        if (column_index == RANGE_IMAGE_WIDTH)
            column_index--;
        const int row_base = (unorganized_point.laser_index % RANGE_IMAGE_HEIGHT) * RANGE_IMAGE_WIDTH;
        if (shift_cell_if_already_occupied)
        {
            if (!std::isnan(organized_points[row_base + column_index].x))
            {
                int right_column_index = column_index + 1;
                if (right_column_index < RANGE_IMAGE_WIDTH && std::isnan(organized_points[row_base + right_column_index].x))
                    column_index = right_column_index;
                else
                {
                    int left_column_index = column_index - 1;
                    if (left_column_index >= 0 && std::isnan(organized_points[row_base + left_column_index].x))
                        column_index = left_column_index;
                }
            }
        }
        KittiPoint& organized_point = organized_points[row_base + column_index];
        organized_point = unorganized_point;
        organized_point.original_kitti_index = original_index;
    }
    for (int k = 0; k < RANGE_IMAGE_HEIGHT * RANGE_IMAGE_WIDTH; k++)
        cell_source[k] = organized_points[k].original_kitti_index;
    return skipped;
}

// makePseudoFiringFromRangeImageColumn for every column, kitti_demo.cpp:123-159. The range image is given as (points, cell_source).
// Outputs firing-major: xyz [2200][64][3], intensity [2200][64], unique [2200][64], stamps [2200].
void korc_make_firings(int64_t n, const float* pts4, const int32_t* cell_source, uint64_t start_stamp, uint64_t end_stamp, int sequence_index,
                       int frame_index, float* xyz, uint8_t* intensity, uint64_t* unique, uint64_t* stamps)
{
    std::vector<KittiPoint> range_image(RANGE_IMAGE_HEIGHT * RANGE_IMAGE_WIDTH, KittiPoint());
    for (int k = 0; k < RANGE_IMAGE_HEIGHT * RANGE_IMAGE_WIDTH; k++)
        if (cell_source[k] >= 0 && cell_source[k] < n)
        {
            const float* p = pts4 + static_cast<int64_t>(cell_source[k]) * 4;
            range_image[k].x = p[0];
            range_image[k].y = p[1];
            range_image[k].z = p[2];
            range_image[k].i = p[3];
            range_image[k].original_kitti_index = cell_source[k];
        }
    for (int column_index = 0; column_index < RANGE_IMAGE_WIDTH; column_index++)
    {
        double elapsed_ratio = static_cast<double>(column_index) / (RANGE_IMAGE_WIDTH - 1);
        double elapsed_time = static_cast<double>(end_stamp - start_stamp) * elapsed_ratio;
        const uint64_t stamp = start_stamp + static_cast<uint64_t>(elapsed_time);
        stamps[column_index] = stamp;
        for (int row_index = 0; row_index < RANGE_IMAGE_HEIGHT; row_index++)
        {
            uint32_t flattened_index = RANGE_IMAGE_WIDTH * row_index + column_index;
            const KittiPoint& kitti_point = range_image[flattened_index];
            const size_t o = static_cast<size_t>(column_index) * RANGE_IMAGE_HEIGHT + row_index;
            xyz[o * 3 + 0] = kitti_point.x;
            xyz[o * 3 + 1] = kitti_point.y;
            xyz[o * 3 + 2] = kitti_point.z;
            const float scaled = kitti_point.i * 255;
            int32_t as_int; // cvttss2si
            if (scaled > -2147483648.f && scaled < 2147483648.f)
                as_int = static_cast<int32_t>(scaled);
            else
                as_int = std::numeric_limits<int32_t>::min();
            intensity[o] = static_cast<uint8_t>(as_int & 0xff);
            unique[o] = (static_cast<uint64_t>(sequence_index) << 48) | (static_cast<uint64_t>(frame_index) << 32) |
                        static_cast<uint64_t>(kitti_point.original_kitti_index);
        }
    }
}

void korc_interpolate(int64_t n_poses, const uint64_t* stamps, const double* poses, uint64_t stamp, double* out12)
{
    to12(interpolate(make_poses(n_poses, stamps, poses), stamp), out12);
}

// getStartEndTimestampsVelodyne, kitti_loader.cpp:525-540
void korc_start_end_stamps(int64_t n, const uint64_t* timestamps_middle, uint64_t* timestamps_start, uint64_t* timestamps_end)
{
    for (int64_t i = 0; i < n - 1; i++)
    {
        timestamps_end[i] = (timestamps_middle[i] + timestamps_middle[i + 1]) / 2;
        timestamps_start[i + 1] = timestamps_end[i];
    }
    timestamps_start[0] = timestamps_middle[0] - 50000000UL;
    timestamps_end[n - 1] = timestamps_middle[n - 1] + 50000000UL;
}

// one line of poses.txt, getAllDynamicTransforms(poses), kitti_loader.cpp:330-369
void korc_pose_from_line(const double* v, const double* cam0_from_x12, double* out12)
{
    Isometry tf_odom_from_first_cam0;
    const double perm[3][3] = {{0, 0, 1}, {-1, 0, 0}, {0, -1, 0}};
    for (int r = 0; r < 3; r++)
    {
        for (int c = 0; c < 3; c++)
            tf_odom_from_first_cam0.linear.a[r][c] = perm[r][c];
        tf_odom_from_first_cam0.translation.v[r] = 0;
    }
    Isometry tf_first_cam0_from_cam0;
    tf_first_cam0_from_cam0.linear = Mat3{{{v[0], v[1], v[2]}, {v[4], v[5], v[6]}, {v[8], v[9], v[10]}}};
    tf_first_cam0_from_cam0.translation = Vec3{{v[3], v[7], v[11]}};
    to12(compose(compose(tf_odom_from_first_cam0, tf_first_cam0_from_cam0), from12(cam0_from_x12)), out12);
}

} // extern "C"
