#!/bin/bash
# PMC passes over the full-chip weight-gradient micro-benchmark (tools/bench_wgrad3.py batch <shape index>): instruction mix, LDS
# conflicts, wait classes of wg3_mega_kernel.   usage: tools/pmc_wg3.sh <tag> <shape index>
tag=$1; idx=$2
export TMPDIR=/tmp
cd /root/repo
run() { rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/${tag}_p -o p --output-format csv -- python tools/bench_wgrad3.py batch $idx > gpurun_out/${tag}_p.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/${tag}_p/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if "wg3_mega" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print("%-28s %16.0f per launch (%d launches)" % (k, acc[k] / max(n[k], 1), n[k]))
PY
  rm -rf gpurun_out/${tag}_p; }
run SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY
run SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
