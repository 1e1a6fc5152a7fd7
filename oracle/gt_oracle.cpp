// gt_oracle.cpp — TEST INFRASTRUCTURE (not shipped, not on the product path).
//
// CPU restatement of KittiEvaluation::generateEuclideanClusteringLabels (src/evaluation/kitti_evaluation.cpp:224-275) including the
// part that lives in a third-party dependency absent from /root/reference and from this image: PCL (libpcl-dev 1.12.1 on the
// reference's Ubuntu 22.04) pcl::ConditionalEuclideanClustering<PointXYZINormal>::segment
// (segmentation/include/pcl/segmentation/impl/conditional_euclidean_clustering.hpp), restated from its published algorithm:
//
//   processed[] = false
//   for every point i in index order, not processed:
//       current_cluster = [i]; processed[i] = true
//       for cii over the growing current_cluster:
//           radiusSearch(current_cluster[cii], cluster_tolerance) -> neighbours (the first, i.e. the query itself, is skipped)
//           every neighbour that is not processed and satisfies condition(seed, neighbour, squared distance) joins and is marked
//       keep the cluster when min_cluster_size <= size <= max_cluster_size
//
// The radius search is a KdTreeFLANN with flann::L2_Simple<float>: squared distance ((dx*dx) + dy*dy) + dz*dz in float; here a
// 1-m grid finds the candidates (the kd-tree itself is not restated; the accepted set is what matters and the condition's strict
// "< 1" is tighter than the search radius).
// ** PARITY UNPINNED **: PCL is not available, the reference's tests hold no vectors for this function. Known difference: with exactly
// coincident points PCL's "skip the first search result" may skip the twin instead of the query; that can only change a cluster whose
// members all coincide.
#include <cmath>
#include <cstdint>
#include <unordered_map>
#include <vector>

namespace
{
const float MAX_DISTANCE = 1.0;      // kitti_evaluation.hpp:55
const int MIN_CLUSTER_SIZE = 10;     // :56
const int MAX_CLUSTER_SIZE = 300000; // :57

struct PointXYZINormal
{
    float x, y, z, intensity, curvature;
};

bool isSameCluster(const PointXYZINormal& p1, const PointXYZINormal& p2, float sqr_dist) // kitti_evaluation.cpp:270-275
{
    return sqr_dist < MAX_DISTANCE * MAX_DISTANCE && p1.curvature == p2.curvature && p1.intensity == p2.intensity;
}

struct CellKey
{
    int x, y, z;
    bool operator==(const CellKey& o) const
    {
        return x == o.x && y == o.y && z == o.z;
    }
};
struct CellHash
{
    size_t operator()(const CellKey& k) const
    {
        return (static_cast<size_t>(static_cast<uint32_t>(k.x)) * 73856093u) ^ (static_cast<size_t>(static_cast<uint32_t>(k.y)) * 19349663u) ^
               (static_cast<size_t>(static_cast<uint32_t>(k.z)) * 83492791u);
    }
};
} // namespace

extern "C" void orc_generate_euclidean_labels(int64_t n, const float* pts4, const uint16_t* semantic, const uint16_t* instance, uint16_t* generated_labels,
                                              int32_t* num_clusters)
{
    // convert to "pcl point cloud" (:227-238)
    std::vector<PointXYZINormal> cloud(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++)
        cloud[i] = PointXYZINormal{pts4[4 * i], pts4[4 * i + 1], pts4[4 * i + 2], static_cast<float>(semantic[i]), static_cast<float>(instance[i])};

    // search structure
    std::unordered_map<CellKey, std::vector<int>, CellHash> grid;
    auto finite = [](const PointXYZINormal& p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z) && std::fabs(p.x) < 1.0e6f &&
                                                        std::fabs(p.y) < 1.0e6f && std::fabs(p.z) < 1.0e6f; };
    auto cell = [](const PointXYZINormal& p) { return CellKey{static_cast<int>(std::floor(p.x)), static_cast<int>(std::floor(p.y)), static_cast<int>(std::floor(p.z))}; };
    for (int64_t i = 0; i < n; i++)
        if (finite(cloud[i]))
            grid[cell(cloud[i])].push_back(static_cast<int>(i));
    const float radius_sqr = MAX_DISTANCE * MAX_DISTANCE;
    std::vector<int> nn_indices;
    std::vector<float> nn_distances;
    auto radiusSearch = [&](int query)
    {
        nn_indices.clear();
        nn_distances.clear();
        const PointXYZINormal& a = cloud[query];
        if (!finite(a))
            return;
        const CellKey c = cell(a);
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dz = -1; dz <= 1; dz++)
                {
                    auto it = grid.find(CellKey{c.x + dx, c.y + dy, c.z + dz});
                    if (it == grid.end())
                        continue;
                    for (int idx : it->second)
                    {
                        if (idx == query)
                            continue; // "nii = 1": the query point itself
                        const PointXYZINormal& b = cloud[idx];
                        float result = 0;
                        float diff = a.x - b.x;
                        result += diff * diff;
                        diff = a.y - b.y;
                        result += diff * diff;
                        diff = a.z - b.z;
                        result += diff * diff;
                        if (result <= radius_sqr)
                        {
                            nn_indices.push_back(idx);
                            nn_distances.push_back(result);
                        }
                    }
                }
    };

    // ConditionalEuclideanClustering::segment
    std::vector<std::vector<int>> points_per_cluster;
    std::vector<bool> processed(static_cast<size_t>(n), false);
    for (int64_t iindex = 0; iindex < n; iindex++)
    {
        if (processed[iindex])
            continue;
        std::vector<int> current_cluster;
        size_t cii = 0;
        current_cluster.push_back(static_cast<int>(iindex));
        processed[iindex] = true;
        while (cii < current_cluster.size())
        {
            radiusSearch(current_cluster[cii]);
            for (size_t nii = 0; nii < nn_indices.size(); ++nii)
            {
                if (processed[nn_indices[nii]])
                    continue;
                if (isSameCluster(cloud[current_cluster[cii]], cloud[nn_indices[nii]], nn_distances[nii]))
                {
                    current_cluster.push_back(nn_indices[nii]);
                    processed[nn_indices[nii]] = true;
                }
            }
            cii++;
        }
        if (static_cast<int>(current_cluster.size()) >= MIN_CLUSTER_SIZE && static_cast<int>(current_cluster.size()) <= MAX_CLUSTER_SIZE)
            points_per_cluster.push_back(current_cluster);
    }

    // generate labels from clustering result (:248-265)
    for (int64_t i = 0; i < n; i++)
        generated_labels[i] = 0;
    uint16_t cluster_index = 1;
    for (const auto& points_in_same_cluster : points_per_cluster)
    {
        for (const auto& point_idx : points_in_same_cluster)
        {
            uint16_t s = semantic[point_idx];
            if (s == 60 || s == 40 || s == 44 || s == 48 || s == 49 || s == 72 || s == 0) // lane-marking road parking sidewalk other-ground terrain unlabeled
                generated_labels[point_idx] = 0;
            else
                generated_labels[point_idx] = cluster_index;
        }
        cluster_index++;
    }
    if (num_clusters)
        *num_clusters = static_cast<int32_t>(points_per_cluster.size());
}
