#!/bin/bash
# Instruction-cache and wait counters of wg3_mega_kernel inside the real train step (the flush packs every fragment block / tile-loop
# instance into one launch, so neighbouring workgroups run different code).   usage: tools/pmc_wg3_step.sh <tag>   env: CGEN_LIB
tag=$1
export TMPDIR=/tmp
cd /root/repo
run() { rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/${tag}_s -o s --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu --no-f32 --no-extra --no-cf > gpurun_out/${tag}_s.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/${tag}_s/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if "wg3_mega" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print("%-28s %16.0f per launch (%d launches)" % (k, acc[k] / max(n[k], 1), n[k]))
PY
  rm -rf gpurun_out/${tag}_s; }
run SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH
run SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA
