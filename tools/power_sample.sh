#!/bin/bash
# samples socket power / clocks while the default bench runs (read-only rocm-smi queries)
python3 bench.py --no-alt --steps 800 --warmup 3 --no-cpu-baseline --no-parity --no-kernel-timer > gpurun_out/power_bench.json 2>/dev/null &
pid=$!
for i in $(seq 1 45); do
  echo "t=$i $(/opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed -E 's/GPU\[0\]\s*: //' | tr '\n' ';')"
  sleep 1
done > gpurun_out/power_samples.txt
wait $pid
tail -c 200 gpurun_out/power_bench.json; echo; cat gpurun_out/power_samples.txt | cut -c1-160
