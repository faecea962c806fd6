#!/bin/bash
# libbifold_hip_probe.so = the product library with gemm.hip compiled under -DP8_PROBE (in-kernel clock stamps)
set -e
cd "$(dirname "$0")/../bifold_amd/csrc"
make -j6 >/dev/null
mkdir -p build_probe
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -DP8_PROBE ${PROBE_FLAGS} -c gemm.hip -o build_probe/gemm.o
objs=$(ls build/*.o | grep -v build/gemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_probe/gemm.o -o ../lib/libbifold_hip_probe.so
echo built ../lib/libbifold_hip_probe.so
