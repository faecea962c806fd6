#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags for gemm.hip>"  ->  bifold_amd/lib/libbifold_hip_<name>.so
# (the product library with gemm.hip recompiled under extra -D flags: A/B builds selected with BIFOLD_HIP_LIB)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../bifold_amd/csrc"
make -j6 >/dev/null
mkdir -p build_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $flags -c gemm.hip -o build_$name/gemm.o
objs=$(ls build/*.o | grep -v build/gemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_$name/gemm.o -o ../lib/libbifold_hip_$name.so
echo built bifold_amd/lib/libbifold_hip_$name.so
