#!/bin/bash
# Build a variant of librdoom_hip.so into _variants/NAME.so with extra compiler flags for ONE kernel source:
#   tools/variant.sh NAME raster "-DFOO=1"
# (A/B on the GPU box: cp _variants/NAME.so rust-doom_amd/librdoom_hip.so between runs.)
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; flags=$3
mkdir -p _variants
obj=_variants/$name.$unit.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude \
  -Wno-unused-function -Wno-bitwise-instead-of-logical $flags -c rust-doom_amd/csrc/hip/$unit.hip -o $obj
objs=$(ls rust-doom_amd/csrc/_obj/*.o | grep -v "/$unit.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/$name.so $obj $objs
echo _variants/$name.so
