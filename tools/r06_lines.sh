#!/bin/bash
# round 6: config A attention with the pipelined forward forced for short sequences, then bench lines of the configurations
set -o pipefail
out=gpurun_out/r06; mkdir -p $out
echo "== clip shapes, default" | tee $out/clip_pipe_ab.txt
python tools/attn_bench.py --dtype fp16 2>&1 | grep -v amdgpu.ids | tee -a $out/clip_pipe_ab.txt
echo "== clip shapes, default" | tee -a $out/clip_pipe_ab.txt
python tools/attn_bench.py --dtype fp16 2>&1 | grep -v amdgpu.ids | tee -a $out/clip_pipe_ab.txt
timeout -k 10 500 python3 bench.py --no-alt --config siglip_seq --batch 128 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_line_siglip_seq_frozen.json 2> $out/siglip.err || { echo siglip failed; tail -5 $out/siglip.err; }
python3 tools/benchline.py $out/bench_line_siglip_seq_frozen.json
timeout -k 10 400 python3 bench.py --no-alt --config vitl14 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_line_vitl14.json 2> /dev/null
python3 tools/benchline.py $out/bench_line_vitl14.json
timeout -k 10 300 python3 bench.py --no-alt --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_line_a.json 2> /dev/null
python3 tools/benchline.py $out/bench_line_a.json
