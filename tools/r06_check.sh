#!/bin/bash
# round 6: the GPU suite (stop at the first failure), tail of its output under gpurun_out/r06
mkdir -p gpurun_out/r06
timeout -k 10 1150 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -25 | tee gpurun_out/r06/gpu_suite.txt
