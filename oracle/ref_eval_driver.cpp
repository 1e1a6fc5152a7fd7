// ref_eval_driver.cpp — TEST INFRASTRUCTURE. extern "C" driver around the REFERENCE's own continuous_clustering::KittiEvaluation (compiled from
// /root/reference/src/evaluation/kitti_evaluation.cpp + kitti_loader.cpp where they lie), with the signatures of orc_eval_frame
// (oracle/eval_oracle.cpp) and orc_generate_euclidean_labels (oracle/gt_oracle.cpp), so that tests/test_reference_build.py can diff those
// restatements against the reference itself. Public API only (kitti_evaluation.hpp:61-73): evaluateGroundPoints, evaluateClusters,
// generateEuclideanClusteringLabels.
//
// Built ONLY by oracle/build_ref.sh and only against a real PCL (common, search, kdtree, segmentation) AND a real Eigen3. No stand-in
// headers, ever: without them the recipe stops and the two restatements stay "parity unpinned". Output: oracle/_ref/libeval_ref.so.
#include <continuous_clustering/evaluation/kitti_evaluation.hpp>

#include <cstdint>
#include <vector>

#include "../include/cc_hip.h"

using namespace continuous_clustering;

extern "C" {

// kitti_evaluation.cpp:44-146
int ref_eval_frame(int64_t n, const uint16_t* semantic, const uint32_t* euclid, const uint8_t* is_ground, const uint32_t* detection,
                   cc_eval_frame_result* r)
{
    std::vector<KittiSegmentationEvaluationPoint> cloud(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++)
    {
        cloud[i].point.semantic_label = semantic[i];
        cloud[i].is_ground_point = is_ground[i] != 0;
        cloud[i].euclidean_clustering_label = euclid[i];
        cloud[i].detection_label = detection[i];
        cloud[i].has_corresponding_point_in_detection_point_cloud = true;
    }
    KittiEvaluation evaluation;
    EvaluationResultForFrame res;
    evaluation.evaluateGroundPoints(cloud, res);
    KittiEvaluation::evaluateClusters(cloud, res);
    r->tp = res.tp;
    r->fn = res.fn;
    r->fp = res.fp;
    r->tn = res.tn;
    r->over_segmentation_entropy = res.over_segmentation_entropy;
    r->under_segmentation_entropy = res.under_segmentation_entropy;
    return 0;
}

// kitti_evaluation.cpp:224-275
void ref_generate_euclidean_labels(int64_t n, const float* pts4, const uint16_t* semantic, const uint16_t* instance, uint16_t* generated_labels)
{
    std::vector<KittiPoint> points(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++)
    {
        points[i].x = pts4[4 * i + 0];
        points[i].y = pts4[4 * i + 1];
        points[i].z = pts4[4 * i + 2];
        points[i].i = pts4[4 * i + 3];
        points[i].semantic_label = semantic[i];
        points[i].instance_label = instance[i];
    }
    KittiEvaluation evaluation;
    const std::vector<uint16_t> labels = evaluation.generateEuclideanClusteringLabels(points);
    for (int64_t i = 0; i < n && i < static_cast<int64_t>(labels.size()); i++)
        generated_labels[i] = labels[i];
}

} // extern "C"
