#!/usr/bin/env python3
"""Per-kernel MFMA-pipe utilisation from ONE rocprofv3 PMC pass (--kernel-trace --output-format csv --pmc
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE).

Units (MI355X_MICROARCH.md, cycle-constants table and DVFS note): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles
summed over the chip's SIMDs (16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16); GRBM_GUI_ACTIVE is the sum over
the 8 XCDs of the cycles the dispatch was active, so GUI/8 = kernel cycles and GUI/8/time = the effective clock;
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave.
    mfma_busy = MFMA_BUSY / (GUI/8 * 1024 SIMDs)         fraction of the matrix pipes' cycles that issued MFMAs
    of_peak   = mfma_busy * clock / 2.4 GHz              the same against the 2.5 PFLOP/s datasheet peak
usage: mfma_summary.py <pmc_dir>"""
import csv
import glob
import sys
from collections import defaultdict

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import short  # noqa: E402

SIMDS = 256 * 4


def main(d):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not f:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    dur = defaultdict(lambda: [0, 0.0])
    t = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    for r in csv.DictReader(open(t[0])):
        k = short(r["Kernel_Name"])
        dur[k][0] += 1
        dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    print(f"{'kernel':46s} {'launches':>8} {'avg us*':>8} {'clock GHz':>9} {'mfma busy':>9} {'of 2.5PF':>8} "
          f"{'wave wait':>9} {'issue stall':>11} {'active':>7}")
    rows = []
    for k, c in per.items():
        n = len(calls[k])
        us = dur[k][1] / max(dur[k][0], 1)
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0 / n
        if cyc <= 0 or us <= 0:
            continue
        clock = cyc / us / 1e3
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (cyc * SIMDS)
        wc = max(c["SQ_WAVE_CYCLES"], 1.0)
        rows.append((n * us, k, n, us, clock, busy, busy * clock / 2.4, c["SQ_WAIT_ANY"] / wc,
                     c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc))
    for tot, k, n, us, clock, busy, peak, w, st, act in sorted(rows, reverse=True)[:18]:
        print(f"{k:46s} {n:8d} {us:8.1f} {clock:9.2f} {100 * busy:8.1f}% {100 * peak:7.1f}% {100 * w:8.1f}% "
              f"{100 * st:10.1f}% {100 * act:6.1f}%")
    print("# * durations of the counter-collection run (kernels serialised).  wave wait / issue stall / active = "
          "SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint shares of a wave's life)")


if __name__ == "__main__":
    main(sys.argv[1])
