#!/bin/bash
# tools/kernel_resources.sh <file.hip> [pattern]: registers / spills / scratch per kernel of one translation unit
# (hipcc -Rpass-analysis=kernel-resource-usage condensed to one line per kernel)
cd "$(dirname "$0")/../bifold_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $EXTRA -c "$1" -o /tmp/kr_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys, subprocess
pat = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None: cur[k] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n)
    if pat and not re.search(pat, n): continue
    print("%-90s v%4s a%3s s%4s spill v%3s s%3s scratch %4s occ %s" % (n[:90], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
' "$2"
rm -f /tmp/kr_$$.o
