#!/bin/bash
# build_ref.sh — TEST INFRASTRUCTURE. Compiles the REFERENCE's own clustering core from the sources where they lie
# (/root/reference/src/clustering/continuous_clustering.cpp + its headers) together with oracle/ref_driver.cpp into
# oracle/_ref/libcc_ref.so, so that tests/test_reference_build.py can diff oracle/cc_oracle.cpp against the reference itself.
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
