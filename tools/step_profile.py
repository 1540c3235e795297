#!/usr/bin/env python3
"""Table of one training step from bench.py's --profile-json dump (dev tool): every conv entry-point call with its geometry,
the kernel instantiation that ran, accumulate flag, time, TFLOP/s and algorithmic TB/s; other entry points summed by name.
usage: python tools/step_profile.py prof.json [fwd|dgrad|wgrad|all]"""
import json
import sys
from collections import defaultdict


def main():
    d = json.load(open(sys.argv[1]))
    what = sys.argv[2] if len(sys.argv) > 2 else "all"
    names = [e["name"] for e in d]
    per = len(d)
    for cand in range(20, len(d) // 2 + 1):
        if names[:cand] == names[cand:2 * cand]:
            per = cand
            break
    step = d[:per]
    print(f"{per} entry-point calls per step, {sum(e['ms'] for e in step):.3f} ms summed")
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for e in step:
        a = agg[e["name"]]
        a[0] += 1; a[1] += e["ms"]; a[2] += e["bytes"]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:28s} n {v[0]:3d}  {v[1]:6.3f} ms  {v[2] / 1e9:6.2f} GB  {v[2] / 1e9 / max(v[1], 1e-9):5.2f} TB/s")
    for i, e in enumerate(step):
        if "geom" not in e or (what != "all" and what not in e["name"]):
            continue
        g = e["geom"]
        print(f"[{i:3d}] {e['name'][9:]:9s} {e['ms'] * 1e3:7.1f}us {e['flops'] / e['ms'] / 1e9:5.0f}TF {e['bytes'] / e['ms'] / 1e9:5.2f}TB/s "
              f"{g[1]:3d}x{g[2]:3d}x{g[3]:4d}->{g[4]:3d}x{g[5]:3d}x{g[6]:4d} k{g[7]}s{g[8]} ld {g[10]},{g[11]},{g[12]} acc {e.get('accumulate', '?')} {e.get('kernel', '')}")


if __name__ == "__main__":
    main()
