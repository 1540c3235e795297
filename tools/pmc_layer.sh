#!/bin/bash
# dev tool: SQ counters of one conv_bench layer (run on the GPU box):  tools/pmc_layer.sh <layer idx> [fwd|dgrad|wgrad]
# several --pmc passes (counter groups must fit the hardware slots); prints per-kernel sums for the igemm/pw/wgrad kernels
L=${1:-9}; W=${2:-fwd}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
  d=gpurun_out/pmcl_$$; rm -rf $d
  rocprofv3 --pmc $grp -d $d -o p --output-format csv -- python3 tools/conv_bench.py $W --only $L --iters 3 $EXTRA > /dev/null 2>&1
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("igemm", "pw_kernel", "wgrad")):
            continue
        a = agg[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(agg.items()):
    print(f"{k:32s} {v / n:16.0f} per launch ({n} launches)")
PY
  rm -rf $d
done
