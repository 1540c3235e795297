#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV of bench.py (dev tool): per step, the wall time between the first and the last
kernel, the time at least one kernel is running (union), the time two run side by side, and the idle gaps on the busiest queue.
usage: python tools/trace_gaps.py <kernel_trace.csv> [last-N-steps]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "0")) for r in rows))
    # a step starts at each nchw_to_s2d / first kernel of the forward: use the optimizer's last kernel as the delimiter
    marks = [i for i, e in enumerate(ev) if "nchw_to_s2d" in e[2]]
    if len(marks) < nsteps + 1:
        print("not enough steps in the trace", len(marks)); return
    for a, b in zip(marks[-nsteps - 1:-1], marks[-nsteps:]):
        seg = ev[a:b]
        t0, t1 = seg[0][0], max(e[1] for e in seg)
        pts = []
        for s, e, _, _ in seg:
            pts.append((s, 1)); pts.append((e, -1))
        pts.sort()
        busy = both = 0; depth = 0; last = pts[0][0]
        for tt, d in pts:
            if depth >= 1: busy += tt - last
            if depth >= 2: both += tt - last
            depth += d; last = tt
        ksum = sum(e - s for s, e, _, _ in seg)
        byq = defaultdict(list)
        for s, e, n, q in seg: byq[q].append((s, e, n))
        line = f"step: wall {1e-6 * (t1 - t0):6.3f} ms  busy(union) {1e-6 * busy:6.3f}  two-deep {1e-6 * both:6.3f}  kernel sum {1e-6 * ksum:6.3f}  kernels {len(seg)}"
        for q, l in byq.items():
            gaps = [l[i + 1][0] - l[i][1] for i in range(len(l) - 1)]
            line += f" | q{q}: n {len(l)} sum {1e-6 * sum(e - s for s, e, _ in l):.3f}"
        print(line)
    # gap histogram of the last step, all queues merged: idle time of the chip between kernels
    seg = ev[marks[-2]:marks[-1]]
    pts = sorted([(s, 1) for s, e, _, _ in seg] + [(e, -1) for s, e, _, _ in seg])
    depth = 0; last = None; idle = []
    for tt, d in pts:
        if depth == 0 and last is not None and d == 1: idle.append(tt - last)
        depth += d
        if depth == 0: last = tt
    idle.sort()
    n = len(idle)
    if n:
        print(f"chip-idle gaps in the last step: {n}, total {1e-3 * sum(idle):.1f} us, median {1e-3 * idle[n // 2]:.2f} us, p90 {1e-3 * idle[int(n * .9)]:.2f} us, max {1e-3 * idle[-1]:.1f} us")


if __name__ == "__main__":
    main()
