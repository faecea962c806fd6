#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of
the same command, each with --kernel-trace --output-format csv) joined with the kernel durations of
the trace.  Counters are in KiB (MI355X_MICROARCH.md, HBM section); on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads, so a corrected column (2x) is printed next to the raw one.

usage: pmc_summary.py <fetch_dir> <write_dir>"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name.replace("void ", ""))
    m = re.match(r"_Z\d+([a-z0-9_]+kernel)I?(.*)", name)
    if m:
        tail = m.group(2)
        flags = re.findall(r"Lb([01])E", tail)
        tc = "bf16" if "DF16bEv" in tail or re.search(r"EDF16b(Li\d+E)?Ev", tail) else ("f32" if re.search(r"EfE?v|EfLi", tail) else "")
        return f"{m.group(1)}<{','.join(flags)}{',' + tc if tc else ''}>"
    return name[:60]


def load(d, counter):
    per = defaultdict(lambda: [0, 0.0])
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not f:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        per[k][0] += 1
        per[k][1] += float(r["Counter_Value"])
    return per


def durations(d):
    per = defaultdict(lambda: [0, 0.0])
    f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        per[k][0] += 1
        per[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return per


def main(fd, wd):
    fetch, write, dur = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE"), durations(fd)
    print(f"{'kernel':46s} {'launches':>8} {'avg us*':>8} {'read MB':>9} {'read corr':>10} {'write MB':>9} {'TB/s corr':>9}")
    rows = []
    for k, (n, kib) in fetch.items():
        w = write.get(k, [0, 0.0])
        us = dur[k][1] / max(dur[k][0], 1)
        rd = kib * 1024 / n / 1e6
        wr = w[1] * 1024 / max(w[0], 1) / 1e6
        rows.append((n * us, k, n, us, rd, wr))
    for tot, k, n, us, rd, wr in sorted(rows, reverse=True)[:16]:
        print(f"{k:46s} {n:8d} {us:8.1f} {rd:9.1f} {2 * rd:10.1f} {wr:9.1f} {(2 * rd + wr) / us:9.2f}")
    print("# * durations are those of the counter-collection run (kernels serialised, slower than the timed run)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
