#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV (`*_kernel_trace.csv`) of bench.py.

    python3 tools/trace_timeline.py <kernel_trace.csv> [steps_to_skip=2] [steps=2]

The per-kernel stats table says how long kernels run; this says how the step's wall time is covered:
  * wall time of the analysed window, time with >= 1 kernel running, idle time (no kernel on the device),
    time with >= 2 kernels running (streams overlapping);
  * per queue: busy time and kernel count;
  * per kernel name: launches, summed duration, EXCLUSIVE time (alone on the device), and the idle gap that
    FOLLOWS it on its own queue before the next kernel of that queue starts (dependency / launch bubbles);
  * the 25 largest idle gaps with the kernels on either side.
A step boundary is an `adam_kernel` launch (one per optimisation step).
"""
import csv
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name[:90]


def main(path, skip=2, steps=2):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    if len(adam) < skip + steps + 1:
        raise SystemExit(f"only {len(adam)} optimisation steps in the trace")
    lo, hi = adam[skip - 1] + 1 if skip else 0, adam[skip + steps - 1] + 1
    win = rows[lo:hi]
    t0, t1 = win[0][0], max(r[1] for r in win)
    wall = t1 - t0
    # sweep: coverage by concurrency level
    ev = []
    for s, e, *_ in win:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    level, last, cover = 0, t0, defaultdict(int)
    for t, d in ev:
        cover[min(level, 3)] += t - last
        last = t
        level += d
    print(f"# {path}")
    print(f"# window: {steps} steps after skipping {skip}; {len(win)} kernels; wall {wall / 1e6 / steps:.3f} ms/step")
    print(f"#   idle (no kernel running)  {cover[0] / 1e6 / steps:8.3f} ms/step  {100 * cover[0] / wall:5.1f} %")
    print(f"#   exactly one kernel        {cover[1] / 1e6 / steps:8.3f} ms/step  {100 * cover[1] / wall:5.1f} %")
    print(f"#   two kernels               {cover[2] / 1e6 / steps:8.3f} ms/step  {100 * cover[2] / wall:5.1f} %")
    print(f"#   three or more             {cover[3] / 1e6 / steps:8.3f} ms/step  {100 * cover[3] / wall:5.1f} %")
    # per queue
    perq = defaultdict(lambda: [0, 0])
    for s, e, n, q, st in win:
        perq[(q, st)][0] += e - s
        perq[(q, st)][1] += 1
    print("# queue/stream: busy ms/step, kernels/step")
    for k, (b, c) in sorted(perq.items(), key=lambda kv: -kv[1][0]):
        print(f"#   queue {k[0]:>3} stream {k[1]:>3}: {b / 1e6 / steps:8.3f}  {c / steps:7.1f}")
    # exclusive time per kernel: portions of its interval where level == 1
    # (second sweep with identity tracking)
    active, last = {}, t0
    excl = defaultdict(int)
    ev2 = []
    for i, (s, e, *_r) in enumerate(win):
        ev2.append((s, 1, i))
        ev2.append((e, 0, i))
    ev2.sort()
    for t, kind, i in ev2:
        if len(active) == 1:
            excl[win[next(iter(active))][2]] += t - last
        last = t
        if kind:
            active[i] = True
        else:
            active.pop(i, None)
    # gap that follows each kernel on its own queue
    follow = defaultdict(int)
    lastq = {}
    for s, e, n, q, st in win:
        k = (q, st)
        if k in lastq:
            pe, pn = lastq[k]
            if s > pe:
                follow[pn] += s - pe
        lastq[k] = (e, n)
    agg = defaultdict(lambda: [0, 0])
    for s, e, n, *_ in win:
        agg[n][0] += e - s
        agg[n][1] += 1
    print(f"{'ms/step':>8} {'excl':>8} {'gap after':>9} {'calls':>6} {'avg us':>8}  kernel")
    for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"{d / 1e6 / steps:8.3f} {excl[n] / 1e6 / steps:8.3f} {follow[n] / 1e6 / steps:9.3f} {c / steps:6.1f} "
              f"{d / c / 1e3:8.1f}  {short(n)}")
    # largest idle gaps on the device
    gaps = []
    end_so_far, prev = win[0][1], win[0][2]
    for s, e, n, *_ in win[1:]:
        if s > end_so_far:
            gaps.append((s - end_so_far, prev, n))
        if e > end_so_far:
            end_so_far, prev = e, n
    gaps.sort(reverse=True)
    print(f"# device idle gaps: {len(gaps) / steps:.0f} per step, total {sum(g[0] for g in gaps) / 1e6 / steps:.3f} ms/step; largest:")
    for g, a, b in gaps[:25]:
        print(f"#   {g / 1e3:8.1f} us  after {short(a)[:50]:<50} before {short(b)[:50]}")
    # the same gaps by the pair of kernels on either side
    pairs = defaultdict(lambda: [0, 0])
    for g, a, b in gaps:
        k = (short(a)[:44], short(b)[:44])
        pairs[k][0] += g
        pairs[k][1] += 1
    print("# idle gaps by (kernel before, kernel after): us/step, count/step")
    for (a, b), (g, c) in sorted(pairs.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"#   {g / 1e3 / steps:8.1f} {c / steps:5.1f}  {a:<44} -> {b}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2, int(sys.argv[3]) if len(sys.argv) > 3 else 2)
