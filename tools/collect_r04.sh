#!/bin/bash
# Round-4 evidence in ONE GPU call (run on the GPU box from the repo root): tools/collect_r04.sh  ->  gpurun_out/r04/*
# Every step writes its own file as it goes (a long silent call is taken for hung).
set -o pipefail
out=gpurun_out/r04
mkdir -p $out
export TMPDIR=/tmp
L=$PWD/bifold_amd/lib
filt() { grep -v amdgpu.ids; }
step() { echo "[$(date +%H:%M:%S)] $*"; }
step "serial kernel stats fp16 / bf16"
tools/serial_stats.sh r04s_fp16 > $out/serial_fp16.log 2>&1 && cp gpurun_out/prof_r04s_fp16/serial_stats.txt $out/serial_kernel_stats_fp16.txt || { echo FAILED serial fp16; tail -3 $out/serial_fp16.log; }
tools/serial_stats.sh r04s_bf16 --dtype bf16 > $out/serial_bf16.log 2>&1 && cp gpurun_out/prof_r04s_bf16/serial_stats.txt $out/serial_kernel_stats_bf16.txt || { echo FAILED serial bf16; tail -3 $out/serial_bf16.log; }
step "concurrent kernel stats + PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy)"
tools/collect_profiles.sh r04p > $out/collect.log 2>&1 && cp gpurun_out/prof_r04p/stats.txt $out/bench_kernel_stats.txt && cp gpurun_out/prof_r04p/hbm_traffic.txt $out/hbm_traffic_pmc.txt && cp gpurun_out/prof_r04p/mfma.txt $out/mfma_busy_pmc.txt || { echo FAILED collect_profiles; tail -3 $out/collect.log; }
step "GEMM vs the vendor library, fp16 and bf16 (same buffers, rotated order)"
{ echo "# tools/gemm_bench.py --variants 8,9 --blaslt --dtype fp16 --rounds 7: v 8 = persistent kernel, BK = 32 loop (round 3); v 9 = BK = 64 full-line loop (round 4, the default); v99 = hipBLASLt through torch.matmul"; timeout -k 5 400 python3 tools/gemm_bench.py --variants 8,9 --blaslt --dtype fp16 --rounds 7 2>&1 | filt; } > $out/gemm_vs_hipblaslt_fp16.txt
{ echo "# the same in bf16"; timeout -k 5 400 python3 tools/gemm_bench.py --variants 8,9 --blaslt --dtype bf16 --rounds 7 2>&1 | filt; } > $out/gemm_vs_hipblaslt_bf16.txt
step "in-kernel probes of the persistent GEMM (BK 32 loop, BK 64 loop)"
{ BIFOLD_HIP_LIB=$L/libbifold_hip_probe.so timeout -k 5 200 python3 tools/p8_probe3.py 8 fp16 2>&1 | filt; BIFOLD_HIP_LIB=$L/libbifold_hip_probe.so timeout -k 5 200 python3 tools/p8_probe3.py 9 fp16 2>&1 | filt; } > $out/p8_probe.txt
step "attention kernels, isolated"
{ timeout -k 5 150 python3 tools/attn_bench.py --dtype fp16 2>&1 | filt; timeout -k 5 150 python3 tools/attn_bench.py --bwd --dtype fp16 2>&1 | filt; } > $out/attn_bench.txt
step "decoder / loss kernels, isolated"
timeout -k 5 100 python3 tools/decoder_bench.py 2>&1 | filt > $out/decoder_bench.txt
step "epilogue instruction rates"
tools/micro/epi_rate > $out/epi_rate.txt 2>&1
step "write-heavy epilogues with / without start skew"
{ for s in "" "4,80"; do BIFOLD_P8_SKEW=$s timeout -k 5 100 python3 tools/p8_res_bench.py 2>&1 | grep sum; done; } > $out/p8_skew.txt
step "bench lines: default (with cpu baseline + alt), bf16, vitl14, siglip_seq frozen / lora, fp8"
timeout -k 10 400 python3 bench.py --no-alt --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench_line.err
timeout -k 10 300 python3 bench.py --no-alt --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline > $out/bench_line_bf16.json 2> /dev/null
timeout -k 10 400 python3 bench.py --no-alt --config vitl14 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_line_vitl14.json 2> /dev/null
step "  siglip_seq"
timeout -k 10 500 python3 bench.py --no-alt --config siglip_seq --batch 128 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_line_siglip_seq_frozen.json 2> /dev/null
timeout -k 10 500 python3 bench.py --no-alt --config siglip_seq --lora --batch 128 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_line_siglip_seq_lora.json 2> /dev/null
timeout -k 10 300 python3 bench.py --no-alt --config fp8 --batch 512 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_line_fp8.json 2> /dev/null
step "LoRA recompute policies (five runs in one process each)"
timeout -k 10 600 python3 tools/lora_recompute_bench.py none mlp full 2>&1 | grep lora_recompute > $out/lora_recompute.txt
step "step timeline"
tools/trace_run.sh r04t > /dev/null 2>&1 && cp gpurun_out/trace_r04t/timeline.txt $out/step_timeline.txt
step done
for f in bench_line bench_line_bf16 bench_line_vitl14 bench_line_siglip_seq_frozen bench_line_siglip_seq_lora bench_line_fp8; do python3 tools/benchline.py $out/$f.json; done
ls $out
