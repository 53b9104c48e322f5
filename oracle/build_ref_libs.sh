#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY (see oracle/README.md).
#
# Builds the reference's vendored numeric libraries from the sources WHERE THEY
# LIE under /root/reference (nothing is copied into this repo) into
# oracle/_ref/ (git-ignored).  Recipe = SURVEY.md §8(c), with two deliberate
# deviations so the resulting binary is portable to the GPU box and has ONE
# fixed arithmetic path:
#   * no -march=native anywhere (OpenGV's CMake forces it, so OpenGV is compiled
#     by build_ref_shim.sh with plain g++ instead of its CMake);
#   * OpenCV CPU_BASELINE=SSE3, CPU_DISPATCH="" (no AVX2/AVX512 run-time
#     dispatch) -> the universal-intrinsics 128-bit path, the closest native
#     analogue of the shipped wasm simd128 build.
# Idempotent: each stage is skipped when its install marker exists.
set -euo pipefail
REF=${ALVA_REFERENCE_ROOT:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
P="$OUT/prefix"
B="$OUT/build"
J=${ALVA_REF_JOBS:-$(nproc)}
mkdir -p "$P" "$B"
if [ ! -d "$REF/src/libs/opencv" ]; then
  echo "reference tree not present at $REF; nothing to build" >&2
  exit 0
fi

if [ ! -f "$P/lib/libopencv_core.a" ]; then
  cmake -G Ninja -S "$REF/src/libs/opencv" -B "$B/opencv" \
    -DCMAKE_BUILD_TYPE=Release -DCMAKE_INSTALL_PREFIX="$P" \
    -DCMAKE_POLICY_VERSION_MINIMUM=3.5 \
    -DBUILD_LIST=core,imgproc,features2d,flann,video,calib3d \
    -DBUILD_SHARED_LIBS=OFF -DENABLE_PIC=ON -DCMAKE_POSITION_INDEPENDENT_CODE=ON \
    -DCPU_BASELINE=SSE3 -DCPU_DISPATCH= \
    -DBUILD_TESTS=OFF -DBUILD_PERF_TESTS=OFF -DBUILD_EXAMPLES=OFF -DBUILD_opencv_apps=OFF \
    -DBUILD_JAVA=OFF -DBUILD_opencv_python2=OFF -DBUILD_opencv_python3=OFF \
    -DWITH_IPP=OFF -DWITH_ITT=OFF -DWITH_OPENCL=OFF -DWITH_TBB=OFF -DWITH_OPENMP=OFF \
    -DWITH_PTHREADS_PF=OFF \
    -DWITH_LAPACK=OFF -DWITH_EIGEN=OFF -DWITH_PROTOBUF=OFF -DWITH_ADE=OFF -DWITH_QUIRC=OFF \
    -DWITH_FFMPEG=OFF -DWITH_GSTREAMER=OFF -DWITH_V4L=OFF -DWITH_GTK=OFF -DWITH_1394=OFF \
    -DWITH_JPEG=OFF -DWITH_PNG=OFF -DWITH_TIFF=OFF -DWITH_WEBP=OFF -DWITH_OPENEXR=OFF \
    -DWITH_JASPER=OFF -DWITH_OPENJPEG=OFF -DWITH_IMGCODEC_HDR=OFF -DWITH_IMGCODEC_SUNRASTER=OFF \
    -DWITH_IMGCODEC_PXM=OFF -DWITH_IMGCODEC_PFM=OFF \
    -DBUILD_ZLIB=ON > "$B/opencv_configure.log" 2>&1
  ninja -C "$B/opencv" -j "$J" install > "$B/opencv_build.log" 2>&1
fi

# Eigen is header-only and its own CMake is broken in the vendored tree
# (scripts/buildtests.in missing): hand Ceres a tiny config shim instead.
SHIM="$B/eigen_shim"
mkdir -p "$SHIM"
cat > "$SHIM/Eigen3Config.cmake" <<EOS
set(EIGEN3_FOUND TRUE)
set(EIGEN3_INCLUDE_DIR "$REF/src/libs/eigen")
set(EIGEN3_INCLUDE_DIRS "$REF/src/libs/eigen")
set(EIGEN3_VERSION_STRING "3.4.0")
if(NOT TARGET Eigen3::Eigen)
  add_library(Eigen3::Eigen INTERFACE IMPORTED)
  set_target_properties(Eigen3::Eigen PROPERTIES INTERFACE_INCLUDE_DIRECTORIES "$REF/src/libs/eigen")
endif()
EOS
cat > "$SHIM/Eigen3ConfigVersion.cmake" <<'EOS'
set(PACKAGE_VERSION "3.4.0")
set(PACKAGE_VERSION_COMPATIBLE TRUE)
if("${PACKAGE_FIND_VERSION}" VERSION_EQUAL "3.4.0")
  set(PACKAGE_VERSION_EXACT TRUE)
endif()
EOS

if [ ! -f "$P/lib/libceres.a" ]; then
  cmake -G Ninja -S "$REF/src/libs/ceres-solver" -B "$B/ceres" \
    -DCMAKE_BUILD_TYPE=Release -DCMAKE_INSTALL_PREFIX="$P" -DCMAKE_CXX_STANDARD=17 \
    -DCMAKE_POLICY_VERSION_MINIMUM=3.5 -DCMAKE_POSITION_INDEPENDENT_CODE=ON \
    -DBUILD_SHARED_LIBS=OFF -DBUILD_EXAMPLES=OFF -DBUILD_TESTING=OFF -DBUILD_BENCHMARKS=OFF \
    -DMINIGLOG=ON -DGFLAGS=OFF -DSUITESPARSE=OFF -DCXSPARSE=OFF -DLAPACK=OFF \
    -DEIGENSPARSE=ON -DCERES_THREADING_MODEL=NO_THREADS \
    -DEigen3_DIR="$SHIM" > "$B/ceres_configure.log" 2>&1
  ninja -C "$B/ceres" -j "$J" install > "$B/ceres_build.log" 2>&1
fi
echo "reference libs ready under $P"
