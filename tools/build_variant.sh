#!/bin/bash
# usage: tools/build_variant.sh <suffix> <extra hipcc flags...>   -> continuous_clustering_amd/libcc_hip_<suffix>.so (A/B and instrumented builds)
suffix=$1; shift
cd "$(dirname "$0")/../continuous_clustering_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-variable -Wno-unused-but-set-variable -Wno-pass-failed -ldl "$@" \
  -o ../libcc_hip_${suffix}.so cc_engine.hip cc_eval.hip cc_kitti.hip cc_gt_labels.hip
