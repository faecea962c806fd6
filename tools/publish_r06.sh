#!/bin/bash
# Copies what tools/collect_r06.sh left in gpurun_out/r06/ (scratch) into profiles/r06_* (tracked, judged).
set -e
src=gpurun_out/r06
cp_if() { if [ -s "$src/$1" ]; then cp "$src/$1" "profiles/$2" && echo "profiles/$2"; fi; }
cp_if serial_kernel_stats_fp16.txt r06_serial_kernel_stats_fp16.txt
cp_if bench_kernel_stats.txt r06_bench_kernel_stats_v1.txt
cp_if hbm_traffic_pmc.txt r06_hbm_traffic_pmc_v1.txt
cp_if mfma_busy_pmc.txt r06_mfma_busy_pmc_v1.txt
cp_if attn_bench.txt r06_attn_bench.txt
cp_if add_ln_bench.txt r06_add_ln_bench.txt
cp_if step_timeline.txt r06_step_timeline.txt
cp_if siglip_seq_frozen_b128_kernel_stats.txt r06_siglip_seq_frozen_b128_kernel_stats.txt
cp_if siglip_seq_lora_b128_kernel_stats.txt r06_siglip_seq_lora_b128_kernel_stats.txt
cp_if attn_pipe_pmc.txt r06_attn_pipe_pmc.txt
cp_if lora_bench.txt r06_lora_bench.txt
cp_if lora_variants.txt r06_lora_load_depth_variants.txt
cp_if gemm_vs_hipblaslt_fp16.txt r06_gemm_vs_hipblaslt_fp16.txt
cp_if observed_parity.jsonl r06_observed_parity.jsonl
for f in bench_line bench_line_bf16 bench_line_vitl14 bench_line_fp8 bench_line_siglip_seq_frozen bench_line_siglip_seq_lora_ride bench_line_siglip_seq_lora_up_pass bench_line_siglip_seq_lora_gemm_form bench_line_siglip_seq_lora_merged; do
  if [ -s "$src/$f.json" ]; then tail -1 "$src/$f.json" > "profiles/r06_$f.json"; echo "profiles/r06_$f.json"; fi
done
