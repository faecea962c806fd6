#!/bin/bash
# resident backward pair at 4 / 3 / 2 / 1 workgroups per CU (extra dynamic LDS): is it occupancy (latency) or a shared pipe?
for pad in 0 6000 18000 45000; do echo "lds pad $pad"; BIFOLD_ATTN_LDS_PAD=$pad python tools/attn_bench.py --bwd --dtype fp16 --only fusion | grep bwd; done
