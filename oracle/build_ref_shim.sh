#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Compiles, with plain g++ and no -march flag
# (portable to the GPU box's host CPU, one Eigen alignment ABI everywhere):
#   * OpenGV 1.0 sources    (/root/reference/src/libs/opengv/src/**/*.cpp)
#   * AlvaAR slam sources   (/root/reference/src/slam/src/*.cpp minus embind.cpp)
#   * oracle/ref_shim.cpp, ref_shim_map.cpp, ref_shim_relpose.cpp, ref_shim_system.cpp   (our extern "C" marshalling layers)
# Linked with --wrap=gettimeofday so that ref_freeze_clock(1) disables Ceres' wall-clock caps (ref_shim_system.cpp).
# from where they lie, and links them with the static OpenCV/Ceres built by
# build_ref_libs.sh into oracle/_ref/libalva_ref.so.  Objects are cached under
# oracle/_ref/build/obj and only rebuilt when the source is newer.
set -euo pipefail
REF=${ALVA_REFERENCE_ROOT:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
P="$OUT/prefix"
OBJ="$OUT/build/obj"
J=${ALVA_REF_JOBS:-$(nproc)}
if [ ! -d "$REF/src/slam/src" ]; then
  echo "reference tree not present at $REF; keeping prebuilt $OUT/libalva_ref.so" >&2
  exit 0
fi
"$HERE/build_ref_libs.sh"
make -s -C "$HERE" >/dev/null
mkdir -p "$OBJ/opengv" "$OBJ/slam"
L="$REF/src/libs"
INC="-I$REF/src/slam/src -I$P/include/opencv4 -I$L/opencv/modules/highgui/include -I$L/opencv/modules/imgcodecs/include \
 -I$L/opencv/modules/videoio/include -I$L/eigen -I$L/eigen/unsupported -I$L/Sophus -I$L/opengv/include -I$P/include -I$P/include/ceres/internal/miniglog"
CXXFLAGS="-O2 -fPIC -w -DNDEBUG"

compile() { # src obj std
  if [ ! -f "$2" ] || [ "$1" -nt "$2" ]; then
    echo "g++ -std=$3 $CXXFLAGS $INC -c '$1' -o '$2'"
  fi
}
{
  while IFS= read -r f; do
    o="$OBJ/opengv/$(echo "${f#$L/opengv/src/}" | tr '/' '_' | sed 's/\.cpp$/.o/')"
    compile "$f" "$o" c++17
  done < <(find "$L/opengv/src" -name '*.cpp' | sort)
  for f in "$REF"/src/slam/src/*.cpp; do
    b=$(basename "$f" .cpp)
    [ "$b" = embind ] && continue
    std=c++17
    # system.cpp calls duration_cast<> unqualified: needs C++20 ADL on template-ids (SURVEY.md §8c gotcha i)
    [ "$b" = system ] && std=c++20
    compile "$f" "$OBJ/slam/$b.o" $std
  done
  compile "$HERE/ref_shim.cpp" "$OBJ/ref_shim.o" c++17
  compile "$HERE/ref_shim_map.cpp" "$OBJ/ref_shim_map.o" c++17
  compile "$HERE/ref_shim_relpose.cpp" "$OBJ/ref_shim_relpose.o" c++17
  compile "$HERE/ref_shim_system.cpp" "$OBJ/ref_shim_system.o" c++17
  compile "$HERE/ref_shim_plane.cpp" "$OBJ/ref_shim_plane.o" c++17
  # f3's second implementation: the reference's processPlane with its three defects repaired (ref_shim_plane.cpp, ref_plane_patch.sed).
  # The copy is made and edited HERE, outside the repository's tracked files; every edit must hit exactly once.
  PATCHED="$OUT/build/patched"
  mkdir -p "$PATCHED"
  if [ ! -f "$PATCHED/system_plane.cpp" ] || [ "$HERE/ref_plane_patch.sed" -nt "$PATCHED/system_plane.cpp" ] || [ "$REF/src/slam/src/system.cpp" -nt "$PATCHED/system_plane.cpp" ]; then
    sed -f "$HERE/ref_plane_patch.sed" "$REF/src/slam/src/system.cpp" > "$PATCHED/system_plane.cpp"
    for pat in 'pointWorldPos.convertTo(points\[i\], CV_32F);' 'alva_ref_plane_pick(n, indicesPicked);' 'A.col(3).setTo(1.0f);' \
               'copyTo(A.row(i).colRange(0, 3));' 'planeCoefficientsMatrix.col(3).setTo(1.0f);' 'copyTo(planeCoefficientsMatrix.row(i).colRange(0, 3));' \
               'copyTo(planePose.rowRange(0, 3).colRange(0, 3));' 'std::vector<float> distsSorted = dists;' 'const float medianDist = distsSorted\['; do
      [ "$(grep -c "$pat" "$PATCHED/system_plane.cpp")" = 1 ] || { echo "ref_plane_patch.sed: edit '$pat' did not apply exactly once" >&2; exit 1; }
    done
  fi
  compile "$PATCHED/system_plane.cpp" "$OBJ/system_plane_patched.o" "c++20 -DSystem=SystemPlanePatched"
  # the product's host-side map layer over the reference's L1 stages (sys_cpu.cpp): host-logic tests without a GPU
  SLAM="$HERE/../alvaar_amd/csrc/slam"
  mkdir -p "$OBJ/syscpu"
  for f in "$SLAM"/*.cpp; do
    o="$OBJ/syscpu/$(basename "$f" .cpp).o"
    if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find "$SLAM" -name '*.hpp' -newer "$o")" ]; then
      echo "g++ -std=c++17 -O2 -g -fPIC -Wall -DNDEBUG -ffp-contract=off -c '$f' -o '$o'"   # (-g: line numbers for tools/host_profile_cpu.py; same code)
    fi
  done
  if [ ! -f "$OBJ/sys_cpu.o" ] || [ "$HERE/sys_cpu.cpp" -nt "$OBJ/sys_cpu.o" ] || [ -n "$(find "$SLAM" "$HERE/alva_oracle.h" -name '*.h*' -newer "$OBJ/sys_cpu.o")" ]; then
    echo "g++ -std=c++17 $CXXFLAGS $INC -I$HERE -I$HERE/../alvaar_amd/csrc -c '$HERE/sys_cpu.cpp' -o '$OBJ/sys_cpu.o'"
  fi
} > "$OBJ/cmds.txt"
if [ -s "$OBJ/cmds.txt" ]; then
  xargs -P "$J" -I{} bash -c '{}' < "$OBJ/cmds.txt"
fi
g++ -shared -o "$OUT/libalva_ref.so" "$OBJ/ref_shim.o" "$OBJ/ref_shim_map.o" "$OBJ/ref_shim_relpose.o" "$OBJ/ref_shim_system.o" "$OBJ/ref_shim_plane.o" "$OBJ/system_plane_patched.o" "$OBJ/sys_cpu.o" "$OBJ"/syscpu/*.o "$OBJ"/slam/*.o "$OBJ"/opengv/*.o \
  -Wl,--start-group "$P/lib/libopencv_video.a" "$P/lib/libopencv_calib3d.a" "$P/lib/libopencv_features2d.a" \
  "$P/lib/libopencv_flann.a" "$P/lib/libopencv_imgproc.a" "$P/lib/libopencv_core.a" -Wl,--end-group \
  "$P/lib/libceres.a" "$P"/lib/opencv4/3rdparty/libzlib.a -lpthread -ldl -Wl,--exclude-libs,ALL -Wl,--wrap=gettimeofday -L"$HERE" -lalva_oracle -Wl,-rpath,'$ORIGIN/..'
echo "built $OUT/libalva_ref.so"
# ref_run: the same objects as a stand-alone program (tests/ref_runner.py: one reproducible run of the reference's System per process)
if [ ! -f "$OUT/ref_run" ] || [ "$HERE/ref_run.cpp" -nt "$OUT/ref_run" ] || [ "$OUT/libalva_ref.so" -nt "$OUT/ref_run" ]; then
  g++ -std=c++17 -O2 -Wall -c "$HERE/ref_run.cpp" -o "$OBJ/ref_run.o"
  g++ -o "$OUT/ref_run" "$OBJ/ref_run.o" "$OBJ/ref_shim.o" "$OBJ/ref_shim_map.o" "$OBJ/ref_shim_relpose.o" "$OBJ/ref_shim_system.o" "$OBJ/ref_shim_plane.o" "$OBJ/system_plane_patched.o" "$OBJ/sys_cpu.o" "$OBJ"/syscpu/*.o "$OBJ"/slam/*.o "$OBJ"/opengv/*.o \
    -Wl,--start-group "$P/lib/libopencv_video.a" "$P/lib/libopencv_calib3d.a" "$P/lib/libopencv_features2d.a" \
    "$P/lib/libopencv_flann.a" "$P/lib/libopencv_imgproc.a" "$P/lib/libopencv_core.a" -Wl,--end-group \
    "$P/lib/libceres.a" "$P"/lib/opencv4/3rdparty/libzlib.a -lpthread -ldl -Wl,--wrap=gettimeofday -L"$HERE" -lalva_oracle -Wl,-rpath,'$ORIGIN/..'
  echo "built $OUT/ref_run"
fi
