// eval_oracle.cpp — TEST INFRASTRUCTURE. CPU restatement of the label compare of the reference's evaluation library
// (src/evaluation/kitti_evaluation.cpp): evaluateGroundPoints :44-84, evaluateClusters :86-146 (nested std::map buckets, entropy
// terms accumulated in ascending key order), calculateMeanAndStdDev :277-293.
// PARITY UNPINNED with respect to the reference binary (kitti_evaluation.cpp includes PCL headers that are not in this image);
// the arithmetic is plain integer counting plus std::log on the host.
#include <cmath>
#include <cstdint>
#include <map>
#include <vector>

#include "../include/cc_hip.h"

extern "C" {

int orc_eval_frame(int64_t n, const uint16_t* semantic, const uint32_t* euclid, const uint8_t* is_ground, const uint32_t* detection,
                   cc_eval_frame_result* r)
{
    *r = cc_eval_frame_result{};
    // label ids: kitti_loader.cpp:566-604 via kitti_evaluation.cpp:12-27
    const uint16_t unlabeled = 0, road = 40, parking = 44, sidewalk = 48, other_ground = 49, lane_marking = 60, terrain = 72;
    for (int64_t i = 0; i < n; i++)
    {
        if (semantic[i] == unlabeled)
            continue;
        const uint16_t s = semantic[i];
        bool gt = s == lane_marking || s == road || s == parking || s == sidewalk || s == other_ground || s == terrain;
        bool seg = is_ground[i] != 0;
        if (gt)
        {
            if (seg)
                r->tp++;
            else
                r->fn++;
        }
        else
        {
            if (seg)
                r->fp++;
            else
                r->tn++;
        }
    }
    std::map<uint32_t, std::vector<int64_t>> by_gt, by_det;
    for (int64_t i = 0; i < n; i++)
    {
        if (euclid[i] != 0)
            by_gt[euclid[i]].push_back(i);
        if (detection[i] != 0)
            by_det[detection[i]].push_back(i);
    }
    for (const auto& g : by_gt)
    {
        std::map<uint32_t, size_t> m;
        for (int64_t i : g.second)
            m[detection[i]]++;
        for (const auto& d : m)
        {
            double frac = static_cast<double>(d.second) / static_cast<double>(g.second.size());
            r->over_segmentation_entropy -= frac * std::log(frac);
        }
    }
    for (const auto& d : by_det)
    {
        std::map<uint32_t, size_t> m;
        for (int64_t i : d.second)
            m[euclid[i]]++;
        if (m.size() == 1 && m.begin()->first == 0)
            continue;
        for (const auto& g : m)
        {
            double frac = static_cast<double>(g.second) / static_cast<double>(d.second.size());
            r->under_segmentation_entropy -= frac * std::log(frac);
        }
    }
    return 0;
}

void orc_eval_mean_std(const double* data, int64_t n, double* mean, double* std_dev)
{
    double m = 0;
    for (int64_t i = 0; i < n; i++)
        m += data[i];
    m /= static_cast<double>(n);
    double s = 0;
    for (int64_t i = 0; i < n; i++)
    {
        double diff = data[i] - m;
        s += diff * diff;
    }
    *mean = m;
    *std_dev = std::sqrt(s / static_cast<double>(n));
}

} // extern "C"
