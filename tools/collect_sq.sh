#!/bin/bash
# SQ / LDS counters of every kernel of one training step (run on the GPU box through gpurun, from the repo root):
#   tools/collect_sq.sh r4 [workload]
# Separate rocprofv3 --pmc passes (8 SQ slots per pass; no trace domains beside --pmc) over the eagerly issued single-stream step
# (a kernel's counters are its own there), summed per kernel name into gpurun_out/<tag>_sq_<wl>.json; tools/sq_summary.py turns that
# into the table under profiles/ (MFMA-busy share, LDS-active share, VALU per MFMA, wait shares).
tag=${1:-r4}; wl=${2:-cfg2}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/${tag}_sq_${wl}.json
echo "{}" > $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  d=gpurun_out/${tag}_sqpass_$i; rm -rf $d
  rocprofv3 --pmc $grp -d $d -o p --output-format csv -- \
      python3 bench.py --workload $wl --eager --no-overlap --no-cpu-baseline --no-roofline --no-parity-leg --steps 3 --warmup 2 > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  python3 - "$d" "$out" <<'PY'
import csv, glob, json, sys, collections
agg = json.load(open(sys.argv[2]))
n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        a = agg.setdefault(k, {})
        c = a.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
        n += 1
json.dump(agg, open(sys.argv[2], "w"))
print("pass rows:", n, "kernels:", len(agg))
PY
  rm -rf $d
done
echo "wrote $out"
