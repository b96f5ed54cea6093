#!/bin/bash
# Branch / wait counters of the fused Block kernel in its micro-benchmark (tools/bench_blk.py).  usage: tools/pmc_blk.sh <tag> [BLK_ONLY side]
tag=$1; side=${2:-48}
export TMPDIR=/tmp
cd /root/repo
BLK_ONLY=$side rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES -d gpurun_out/${tag}_b -o b --output-format csv -- python tools/bench_blk.py 3 > gpurun_out/${tag}_b.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/${tag}_b/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void cgen::", "").split("(")[0]
    if "blk3" in k or "conv_px" in k or "conv_ws" in k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    a = {c: acc[k][c] / max(n[k][c], 1) for c in acc[k]}
    w = a.get("SQ_WAVES", 1) or 1
    print("%-44s per wave: branch %5.0f salu %5.0f | wave cycles x4 %7.0f: issuing %4.1f %% waiting %4.1f %% waiting for instructions %4.1f %%" % (
        k[:44], a.get("SQ_INSTS_BRANCH", 0) / w, a.get("SQ_INSTS_SALU", 0) / w, a.get("SQ_WAVE_CYCLES", 0) / w,
        100 * a.get("SQ_ACTIVE_INST_ANY", 0) / a.get("SQ_WAVE_CYCLES", 1), 100 * a.get("SQ_WAIT_ANY", 0) / a.get("SQ_WAVE_CYCLES", 1), 100 * a.get("SQ_WAIT_INST_ANY", 0) / a.get("SQ_WAVE_CYCLES", 1)))
PY
rm -rf gpurun_out/${tag}_b
