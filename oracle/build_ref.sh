#!/bin/bash
# build_ref.sh — TEST INFRASTRUCTURE. Compiles the REFERENCE's own sources where they lie, each with a thin extern "C" driver of ours that
# only uses the reference's public API, so that tests/test_reference_build.py can diff the oracle's restatements against the reference itself:
#   oracle/_ref/libcc_ref.so      src/clustering/continuous_clustering.cpp + oracle/ref_driver.cpp         (needs Eigen3)   <-> oracle/cc_oracle.cpp
#   oracle/_ref/libloader_ref.so  src/evaluation/kitti_loader.cpp + oracle/ref_loader_driver.cpp           (needs Eigen3)   <-> oracle/kitti_oracle.cpp
#   oracle/_ref/libeval_ref.so    src/evaluation/kitti_evaluation.cpp + kitti_loader.cpp + oracle/ref_eval_driver.cpp (needs Eigen3 AND PCL)
#                                                                                                          <-> oracle/eval_oracle.cpp, gt_oracle.cpp
#
# The reference core includes <Eigen/Geometry>. This recipe ONLY runs against a real Eigen3 installation (pkg-config eigen3,
# $EIGEN3_INCLUDE_DIR, or one of the usual system locations). It never writes, fetches or substitutes headers: when Eigen3 is not
# installed it prints why and exits 3, and the oracle keeps the label "parity unpinned" (DESIGN.md section 3).
# Exit codes: 0 built, 3 prerequisites missing (reference tree or Eigen3), other = compile error.
set -u
here="$(cd "$(dirname "$0")" && pwd)"
ref="${CC_REFERENCE_ROOT:-/root/reference}"
if [ ! -f "$ref/src/clustering/continuous_clustering.cpp" ]; then
  echo "build_ref: reference tree not found at $ref (it never travels to the GPU box)"; exit 3
fi
eigen=""
if command -v pkg-config >/dev/null 2>&1 && pkg-config --exists eigen3; then
  eigen="$(pkg-config --variable=includedir eigen3)/eigen3"
  [ -f "$eigen/Eigen/Geometry" ] || eigen="$(pkg-config --cflags-only-I eigen3 | sed 's/-I//g' | awk '{print $1}')"
fi
for cand in "${EIGEN3_INCLUDE_DIR:-}" /usr/include/eigen3 /usr/local/include/eigen3 /opt/eigen3/include/eigen3; do
  if [ -z "$eigen" ] && [ -n "$cand" ] && [ -f "$cand/Eigen/Geometry" ] && [ -f "$cand/Eigen/src/Core/util/Macros.h" ]; then eigen="$cand"; fi
done
if [ -z "$eigen" ] || [ ! -f "$eigen/Eigen/src/Core/util/Macros.h" ]; then
  echo "build_ref: no Eigen3 installation found (pkg-config eigen3, \$EIGEN3_INCLUDE_DIR, /usr/include/eigen3 ...): the reference core cannot be built here"; exit 3
fi
echo "build_ref: Eigen3 at $eigen ($(grep -E 'define EIGEN_(WORLD|MAJOR|MINOR)_VERSION' "$eigen/Eigen/src/Core/util/Macros.h" | awk '{print $3}' | paste -sd.))"
mkdir -p "$here/_ref"
# same optimisation level and FP model as the reference's CMakeLists.txt Release build (-O2-class, no -ffast-math); x86-64 SSE2 arithmetic
${CXX:-g++} -O2 -std=c++17 -fPIC -shared -pthread -I"$ref/include" -I"$eigen" \
  -o "$here/_ref/libcc_ref.so" "$ref/src/clustering/continuous_clustering.cpp" "$here/ref_driver.cpp" || exit 1
echo "build_ref: built $here/_ref/libcc_ref.so"
# ---- the KITTI loader (Eigen3 only) -----------------------------------------------------------------------------------------------
${CXX:-g++} -O2 -std=c++17 -fPIC -shared -pthread -I"$ref/include" -I"$eigen" \
  -o "$here/_ref/libloader_ref.so" "$ref/src/evaluation/kitti_loader.cpp" "$here/ref_loader_driver.cpp" || exit 1
echo "build_ref: built $here/_ref/libloader_ref.so"
# ---- the evaluation library (Eigen3 + a real PCL: headers and the libraries kitti_evaluation.cpp links, CMakeLists.txt of the reference) ----
pcl_cflags=""; pcl_libs=""
if command -v pkg-config >/dev/null 2>&1; then
  for v in 1.14 1.13 1.12 1.11 1.10 1.9 1.8; do
    if pkg-config --exists "pcl_segmentation-$v" 2>/dev/null; then
      pcl_cflags="$(pkg-config --cflags pcl_segmentation-$v pcl_search-$v pcl_kdtree-$v pcl_common-$v)"
      pcl_libs="$(pkg-config --libs pcl_segmentation-$v pcl_search-$v pcl_kdtree-$v pcl_common-$v)"
      break
    fi
  done
fi
if [ -z "$pcl_cflags" ]; then
  for cand in "${PCL_INCLUDE_DIR:-}" /usr/include/pcl-1.14 /usr/include/pcl-1.13 /usr/include/pcl-1.12 /usr/include/pcl-1.10 /usr/include/pcl-1.8 /usr/local/include/pcl-1.14; do
    if [ -n "$cand" ] && [ -f "$cand/pcl/point_types.h" ] && [ -f "$cand/pcl/segmentation/conditional_euclidean_clustering.h" ]; then
      pcl_cflags="-I$cand"; pcl_libs="-lpcl_segmentation -lpcl_search -lpcl_kdtree -lpcl_common"; break
    fi
  done
fi
if [ -z "$pcl_cflags" ]; then
  echo "build_ref: no PCL installation found (pkg-config pcl_segmentation-1.x, \$PCL_INCLUDE_DIR, /usr/include/pcl-1.x): libeval_ref.so is not built; oracle/eval_oracle.cpp and gt_oracle.cpp stay parity unpinned"
  exit 0
fi
${CXX:-g++} -O2 -std=c++17 -fPIC -shared -pthread -I"$ref/include" -I"$eigen" $pcl_cflags \
  -o "$here/_ref/libeval_ref.so" "$ref/src/evaluation/kitti_evaluation.cpp" "$ref/src/evaluation/kitti_loader.cpp" "$here/ref_eval_driver.cpp" $pcl_libs || exit 1
echo "build_ref: built $here/_ref/libeval_ref.so"
