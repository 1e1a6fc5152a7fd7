#!/bin/bash
# usage: tools/build_head_variant.sh <suffix> [rev]   -> continuous_clustering_amd/libcc_hip_<suffix>.so built from the csrc/ of a git revision (default HEAD):
# the baseline of a same-box A/B against the working tree's library (tools/ab_lib.sh)
suffix=$1; rev=${2:-HEAD}
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d); mkdir -p $tmp/continuous_clustering_amd/csrc $tmp/include
(cd $root && git archive $rev continuous_clustering_amd/csrc include | tar -x -C $tmp)
cd $tmp/continuous_clustering_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-variable -Wno-unused-but-set-variable -Wno-pass-failed -ldl \
  -o $root/continuous_clustering_amd/libcc_hip_${suffix}.so cc_engine.hip cc_eval.hip cc_kitti.hip cc_gt_labels.hip 2>&1 | grep -i "error" ; rm -rf $tmp
ls -la $root/continuous_clustering_amd/libcc_hip_${suffix}.so
