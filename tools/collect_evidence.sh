#!/bin/bash
# Round evidence beyond collect_profiles.sh (run on the GPU box): in-kernel probes and ablations of the persistent GEMM,
# the vendor-library comparison, the attention backward's workgroup timeline, the step's kernel timeline.
#   tools/collect_evidence.sh <tag>   ->  gpurun_out/evidence_<tag>/*.txt
set -o pipefail
tag=${1:-r02}
out=gpurun_out/evidence_$tag
mkdir -p $out
L=$PWD/bifold_amd/lib
filt() { grep -v amdgpu.ids; }
{ echo "# tools/p8_probe3.py on libbifold_hip_probe.so (gemm.hip built with -DP8_PROBE): workgroup 0's second work item"; BIFOLD_HIP_LIB=$L/libbifold_hip_probe.so timeout -k 5 200 python3 tools/p8_probe3.py 2>&1 | filt; } > $out/p8_probe.txt || exit 1
{ echo "# timing ablation (wrong results): ONE LDS-DMA piece per wave and operand stage instead of two (-DP8_ABL_PIECES=1)"; BIFOLD_HIP_LIB=$L/libbifold_hip_half.so timeout -k 5 200 python3 tools/p8_probe3.py 2>&1 | filt; } > $out/p8_probe_half_dma.txt || exit 1
{ echo "# timing ablation (wrong results): NO LDS-DMA inside the main loop (-DP8_ABL_PIECES=0)"; BIFOLD_HIP_LIB=$L/libbifold_hip_nodma.so timeout -k 5 200 python3 tools/p8_probe3.py 2>&1 | filt; } > $out/p8_probe_no_dma.txt || exit 1
{ echo "# tools/blaslt_ref.py: hipBLASLt (through torch.nn.functional.linear) vs the persistent GEMM, same buffers, isolated"; timeout -k 5 300 python3 tools/blaslt_ref.py 2>&1 | filt; } > $out/gemm_vs_hipblaslt.txt || exit 1
{ echo "# tools/attn_probe.py on libbifold_hip_attnprobe.so (attention.hip built with -DATTN_PROBE): attn_bwd_q_kernel, 275 x 32, B = 256"
  for cfg in "4 1" "5 2" "9 1"; do set -- $cfg; echo "== $1 waves per workgroup, $2 workgroup(s) per (batch, head)"; BIFOLD_HIP_LIB=$L/libbifold_hip_attnprobe.so BIFOLD_ATTN_BWD_WPB=$1 BIFOLD_ATTN_BWD_SPLIT=$2 PROBE_SPLIT=$2 timeout -k 5 100 python3 tools/attn_probe.py 2>&1 | filt; done; } > $out/attn_bwd_probe.txt || exit 1
{ echo "# tools/p8f_probe.py: the persistent e4m3 GEMM (same probe build)"; BIFOLD_HIP_LIB=$L/libbifold_hip_probe.so timeout -k 5 200 python3 tools/p8f_probe.py 2>&1 | filt; echo "# tools/fp8_bench.py: e4m3 vs bf16 persistent GEMM, isolated"; timeout -k 5 200 python3 tools/fp8_bench.py 2>&1 | filt; } > $out/fp8_gemm.txt || exit 1
{ echo "# tools/p8_probe_layouts.py: main-loop cost by operand layout"; BIFOLD_HIP_LIB=$L/libbifold_hip_probe.so timeout -k 5 200 python3 tools/p8_probe_layouts.py 2>&1 | filt; } > $out/p8_probe_layouts.txt || exit 1
{ echo "# tools/attn_bench.py --bwd (isolated, B = 256)"; timeout -k 5 100 python3 tools/attn_bench.py --bwd 2>&1 | filt; } > $out/attn_bench.txt || exit 1
{ echo "# tools/micro/write_bw: streaming 1 GiB, 16 B per lane"; timeout -k 5 100 tools/micro/write_bw; } > $out/hbm_stream_bw.txt || exit 1
tools/trace_run.sh $tag > /dev/null || exit 1
cp gpurun_out/trace_$tag/timeline.txt $out/step_timeline.txt
python3 bench.py --no-alt --steps 6 --warmup 3 --no-cpu-baseline --no-parity --per-shape 2>&1 | filt > $out/per_shape_in_step.txt || exit 1
ls -la $out
