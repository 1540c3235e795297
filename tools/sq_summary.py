#!/usr/bin/env python3
"""Table of the SQ / LDS counters collected by tools/collect_sq.sh, one row per kernel (dev tool, CPU).
usage: python tools/sq_summary.py gpurun_out/r4_sq_cfg2.json [steps_in_run] > profiles/r4_sq_counters.txt

Columns (per launch averages; rocprofv3 sums a counter over the 8 XCDs / all SIMDs):
  cycles      = SQ_BUSY_CYCLES / 32 shader engines: the kernel's duration in shader clocks
  mfma%       = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles): share of the matrix pipes' time that was busy
  lds%        = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles): share of the LDS arrays' time serving ds_ instructions (LDS-DMA writes are not in it)
  bank%       = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  valu/mfma   = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA
  waves       = SQ_WAVE_CYCLES x 4 / (1024 SIMDs x cycles): average resident waves per SIMD (SQ_WAVE_CYCLES counts quad-cycles)
  wait% / stall% / issue% = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES: wave time parked at s_waitcnt or a
                barrier / ready but not issued (pipe busy, dependency) / issuing"""
import json
import sys


def main():
    d = json.load(open(sys.argv[1]))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows = []
    for k, c in d.items():
        g = lambda n: (c[n][0] / max(c[n][1], 1)) if n in c else 0.0
        n_l = max((v[1] for v in c.values()), default=0)
        cyc = g("SQ_BUSY_CYCLES") / 32.0
        if cyc <= 0:
            continue
        wc = g("SQ_WAVE_CYCLES")
        mf = g("SQ_INSTS_MFMA")
        rows.append(dict(k=k, n=n_l / steps, cyc=cyc, tot=cyc * n_l / steps,
                         mfma=100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cyc),
                         lds=100 * g("SQ_LDS_IDX_ACTIVE") / (256 * cyc),
                         bank=100 * g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1),
                         vpm=(g("SQ_INSTS_VALU") - mf) / mf if mf > 0 else float("nan"),
                         waves=wc * 4 / (1024 * cyc),
                         wait=100 * g("SQ_WAIT_ANY") / max(wc, 1), stall=100 * g("SQ_WAIT_INST_ANY") / max(wc, 1),
                         issue=100 * g("SQ_ACTIVE_INST_ANY") / max(wc, 1)))
    rows.sort(key=lambda r: -r["tot"])
    tot = sum(r["tot"] for r in rows)
    print(__doc__.split("Columns")[0].strip().split("\n")[0])
    print("Columns" + __doc__.split("Columns")[1])
    print(f"{'kernel':60s} {'n/step':>6s} {'kcycles':>8s} {'share':>6s} {'mfma%':>6s} {'lds%':>5s} {'bank%':>5s} {'valu/mfma':>9s} {'waves':>5s} {'wait%':>5s} {'stall%':>6s} {'issue%':>6s}")
    for r in rows:
        if r["tot"] / tot < 0.002:
            continue
        print(f"{r['k'][:60]:60s} {r['n']:6.1f} {r['cyc'] / 1e3:8.1f} {100 * r['tot'] / tot:5.1f}% {r['mfma']:6.1f} {r['lds']:5.1f} {r['bank']:5.1f} "
              f"{r['vpm']:9.1f} {r['waves']:5.2f} {r['wait']:5.1f} {r['stall']:6.1f} {r['issue']:6.1f}")


if __name__ == "__main__":
    main()
