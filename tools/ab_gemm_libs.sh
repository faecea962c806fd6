#!/bin/bash
# tools/ab_gemm_libs.sh <name> ...: tools/gemm_bench.py (NT shapes, variant 9 = the library's persistent BK64 path) once
# per library build bifold_amd/lib/libbifold_hip_<name>.so ("default" = the product library), two interleaved passes.
out=gpurun_out/ab_gemm
mkdir -p $out
for rep in 1 2; do
  for name in "$@"; do
    lib=bifold_amd/lib/libbifold_hip_$name.so
    [ "$name" = default ] && lib=bifold_amd/lib/libbifold_hip.so
    BIFOLD_HIP_LIB=$lib timeout -k 10 300 python3 tools/gemm_bench.py --variants 9 --rounds 7 --dtype fp16 --only "${ONLY:-}" > $out/${name}_$rep.txt 2>&1 || { echo "$name failed"; tail -5 $out/${name}_$rep.txt; exit 1; }
    echo "== $name rep $rep"; grep -E "^v 9" $out/${name}_$rep.txt | grep -E " nt |TOTAL" | awk '{printf "%s %s %s us %s TF/s | ", $3, $4, $7, $9} END {print ""}'
  done
done
