#!/bin/bash
# Collect the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r3 [workload]
# 1. default bench line (launch-list replay, two streams, CPU baseline)            -> gpurun_out/<tag>_bench_<wl>.log
# 2. rocprofv3 --kernel-trace --stats of the SAME step issued eagerly on ONE stream -> gpurun_out/<tag>_kt_<wl>/  (a kernel's
#    duration is its own there; 25 steps: --warmup 5 --steps 20, no mode selection, no instrumented pass)
# 3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (7 steps each)       -> gpurun_out/<tag>_pmc{f,w}_<wl>/
# The summaries are copied into profiles/ by hand afterwards (tools/pmc_summary.py for the counters).
tag=${1:-r3}; wl=${2:-cfg2}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py --workload $wl --steps 20 --warmup 5 > gpurun_out/${tag}_bench_${wl}.log 2>&1 || exit 1
tail -c 2500 gpurun_out/${tag}_bench_${wl}.log | head -c 600; echo
rm -rf gpurun_out/${tag}_kt_${wl}
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_kt_${wl} -o kt --output-format csv -- \
    python3 bench.py --workload $wl --eager --no-overlap --no-cpu-baseline --no-roofline --no-parity-leg --steps 20 --warmup 5 > gpurun_out/${tag}_kt_${wl}.log 2>&1 || exit 1
find gpurun_out/${tag}_kt_${wl} -name "*kernel_stats.csv" | head -1
if [ "$wl" = "cfg2" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/${tag}_pmc_${c}_${wl}; rm -rf $d
    rocprofv3 --pmc $c -d $d -o p --output-format csv -- \
        python3 bench.py --workload $wl --eager --no-overlap --no-cpu-baseline --no-roofline --no-parity-leg --steps 5 --warmup 2 > $d.log 2>&1 || exit 1
    # keep only what the summary needs (the raw CSVs are tens of MB)
    python3 - "$d" "$c" <<'PY'
import csv, glob, json, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == sys.argv[2]:
            k = r["Kernel_Name"].split("(")[0]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
json.dump({k: v for k, v in agg.items()}, open(sys.argv[1] + "_summary.json", "w"))
print(sys.argv[2], "kernels:", len(agg))
PY
    rm -rf $d
  done
fi
