#!/bin/bash
# builds a variant of libsffgpu.so beside the shipped one: bash profiles/build_variant.sh <name> "<EXTRA flags>"
# -> space_filling_forest_star_amd/libsffgpu_<name>.so (select it with SFFGPU_LIB=libsffgpu_<name>.so)
set -eu
name=$1; extra=${2:-}
root=$(cd "$(dirname "$0")/.." && pwd)
w=/tmp/sffgpu_variant_$name
rm -rf $w; mkdir -p $w/pkg
cp -r $root/space_filling_forest_star_amd/csrc $w/pkg/csrc
cp -r $root/space_filling_forest_star_amd/cli $w/pkg/cli
ln -s $root/include $w/include
rm -f $w/pkg/csrc/*.o
make -s -j8 -C $w/pkg/csrc EXTRA="$extra" ../libsffgpu.so
cp $w/pkg/libsffgpu.so $root/space_filling_forest_star_amd/libsffgpu_$name.so
echo built libsffgpu_$name.so
