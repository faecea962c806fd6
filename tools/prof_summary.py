#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short text table."""
import csv
import sys


def short(name: str) -> str:
    name = name.replace("void ", "")
    for a, b in (("(bool _Accum", "("), ("_Z", "_Z")):
        pass
    return name[:110]


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# source: {path}")
    print(f"# total kernel time {tot / 1e6:.2f} ms over {steps} steps = {tot / 1e6 / steps:.2f} ms/step")
    print(f"{'ms/step':>9} {'%':>6} {'calls/step':>10} {'avg us':>9}  kernel")
    for r in rows:
        t = float(r["TotalDurationNs"])
        if t / tot < 0.0005:
            continue
        print(f"{t / 1e6 / steps:9.3f} {100 * t / tot:6.2f} {int(r['Calls']) / steps:10.1f} {float(r['AverageNs']) / 1e3:9.1f}  {short(r['Name'])}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
