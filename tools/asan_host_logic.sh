#!/usr/bin/env bash
# The host-side map layer (alvaar_amd/csrc/slam/*.cpp) under AddressSanitizer, without a GPU: the GPU-less harness (oracle/sys_cpu.cpp, test
# infrastructure; needs oracle/_ref built and /root/reference present) with the map layer's sources and the harness compiled -fsanitize=address,
# run over the bench stream in check mode with a System::reset in the middle.  Destroyed map-point objects live in an arena that stays
# mapped: slam.hpp poisons their boxes in sanitizer builds, so a use after destruction is reported like a heap use-after-free.
#   usage: tools/asan_host_logic.sh [frames=500]
set -euo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
REF=${ALVA_REFERENCE_ROOT:-/root/reference}
OBJ=$R/oracle/_ref/build/obj; P=$R/oracle/_ref/prefix; HERE=$R/oracle; L=$REF/src/libs
W=${TMPDIR:-/tmp}/alva_asan; mkdir -p "$W"
INC="-I$REF/src/slam/src -I$P/include/opencv4 -I$L/opencv/modules/highgui/include -I$L/opencv/modules/imgcodecs/include -I$L/opencv/modules/videoio/include -I$L/eigen -I$L/eigen/unsupported -I$L/Sophus -I$L/opengv/include -I$P/include -I$P/include/ceres/internal/miniglog"
F="-std=c++17 -O1 -g -fPIC -fsanitize=address -fno-omit-frame-pointer -DNDEBUG -ffp-contract=off"
for f in "$R"/alvaar_amd/csrc/slam/*.cpp; do g++ $F -c "$f" -o "$W/$(basename "$f" .cpp).o" & done
g++ $F -w $INC -I"$HERE" -I"$R/alvaar_amd/csrc" -c "$HERE/sys_cpu.cpp" -o "$W/sys_cpu.o" &
wait
g++ -shared -fsanitize=address -o "$W/libalva_ref_asan.so" "$OBJ"/ref_shim.o "$OBJ"/ref_shim_map.o "$OBJ"/ref_shim_relpose.o "$OBJ"/ref_shim_system.o "$W"/*.o \
  "$OBJ"/slam/*.o "$OBJ"/opengv/*.o -Wl,--start-group "$P"/lib/libopencv_video.a "$P"/lib/libopencv_calib3d.a "$P"/lib/libopencv_features2d.a \
  "$P"/lib/libopencv_flann.a "$P"/lib/libopencv_imgproc.a "$P"/lib/libopencv_core.a -Wl,--end-group "$P"/lib/libceres.a "$P"/lib/opencv4/3rdparty/libzlib.a \
  -lpthread -ldl -Wl,--exclude-libs,ALL -Wl,--wrap=gettimeofday -L"$HERE" -lalva_oracle -Wl,-rpath,"$HERE"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 ALVA_ASAN_LIB="$W/libalva_ref_asan.so" python "$R/tools/asan_host_logic.py" "${1:-500}"
