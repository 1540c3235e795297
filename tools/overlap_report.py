#!/usr/bin/env python3
"""Dev tool: what the second stream buys.  Compares a rocprofv3 --kernel-trace of the two-stream replayed step with the per-kernel
averages of the single-stream run (kernel_stats.csv): per kernel name the inflation under overlap, per queue the busy time, and the
part of the step during which only ONE queue has a kernel in flight (the serial remainder).
usage: python tools/overlap_report.py <two_stream_kernel_trace.csv> <single_stream_kernel_stats.csv> [last-N-steps]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    alone = {r["Name"].split("(")[0]: float(r["AverageNs"]) for r in csv.DictReader(open(sys.argv[2]))}
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "0")) for r in rows))
    marks = [i for i, e in enumerate(ev) if "nchw_to_s2d" in e[2]]
    seg = ev[marks[-nsteps - 1]:marks[-1]]
    qs = sorted({e[3] for e in seg}, key=lambda q: -sum(e[1] - e[0] for e in seg if e[3] == q))
    print("queues by busy time:", [(q, round(1e-6 * sum(e[1] - e[0] for e in seg if e[3] == q) / nsteps, 3)) for q in qs])
    main_q = qs[0]
    # time with depth-1 on main only / side only / both
    pts = []
    for s, e, n, q in seg:
        k = 0 if q == main_q else 1
        pts.append((s, k, 1)); pts.append((e, k, -1))
    pts.sort()
    d = [0, 0]; last = pts[0][0]; acc = defaultdict(int)
    for t, k, dd in pts:
        acc[(d[0] > 0, d[1] > 0)] += t - last
        d[k] += dd; last = t
    tot = sum(acc.values())
    print(f"per step: wall {1e-6 * tot / nsteps:.3f} ms | main only {1e-6 * acc[(True, False)] / nsteps:.3f} | side only {1e-6 * acc[(False, True)] / nsteps:.3f} "
          f"| both {1e-6 * acc[(True, True)] / nsteps:.3f} | idle {1e-6 * acc[(False, False)] / nsteps:.3f}")
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, n, q in seg:
        a = agg[(n, q == main_q)]
        a[0] += 1; a[1] += e - s
    print(f"{'kernel':70s} {'queue':5s} {'n/step':>6s} {'us two-stream':>13s} {'us alone':>9s} {'ratio':>6s} {'extra us/step':>13s}")
    out = []
    for (n, m), (c, t) in agg.items():
        a = alone.get(n)
        avg = t / c
        out.append(((avg - (a or avg)) * c / nsteps, n, m, c / nsteps, avg, a))
    tot_extra = 0
    for extra, n, m, c, avg, a in sorted(out, key=lambda x: -x[0])[:40]:
        tot_extra += extra
        print(f"{n[:70]:70s} {'main' if m else 'side':5s} {c:6.1f} {avg / 1e3:13.1f} {(a or 0) / 1e3:9.1f} {avg / a if a else 0:6.2f} {extra / 1e3:13.1f}")
    print(f"sum of inflation: {sum(o[0] for o in out) / 1e3:.1f} us/step")


if __name__ == "__main__":
    main()
