# A/B builds of ONE source file with extra -D flags: tools/build_variant.sh <tag> <file.hip> [-DNAME=VALUE ...]
# -> ab_variants/lib_<tag>.so (all other objects are the in-tree build's); run with ALVA_LIB=ab_variants/lib_<tag>.so
set -e
tag=$1; src=$2; shift 2
mkdir -p ab_variants
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function \
  -Iinclude "$@" -c $src -o ab_variants/${base}_$tag.o
objs=$(ls alvaar_amd/csrc/*.o alvaar_amd/csrc/slam/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_variants/lib_$tag.so $objs ab_variants/${base}_$tag.o
echo ab_variants/lib_$tag.so
