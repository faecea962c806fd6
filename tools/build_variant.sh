#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags>" [source.hip (default gemm.hip)]  ->  bifold_amd/lib/libbifold_hip_<name>.so
# (the product library with ONE source recompiled under extra -D flags: A/B builds selected with BIFOLD_HIP_LIB)
set -e
name=$1; flags=$2; src=${3:-gemm.hip}
obj=${src%.hip}.o
cd "$(dirname "$0")/../bifold_amd/csrc"
make -j6 >/dev/null
mkdir -p build_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $flags -c $src -o build_$name/$obj
objs=$(ls build/*.o | grep -v "build/$obj\|amdgcn")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_$name/$obj -o ../lib/libbifold_hip_$name.so
echo built bifold_amd/lib/libbifold_hip_$name.so
