export TMPDIR=/tmp
cd /root/repo
rm -rf gpurun_out/ic
ONLY=7 ITERS=5 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH -d gpurun_out/ic -o ic --output-format csv -- python tools/bench_conv.py bf16 fwd > gpurun_out/ic.log 2>&1
ONLY=7 ITERS=5 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d gpurun_out/ic2 -o ic2 --output-format csv -- python tools/bench_conv.py bf16 fwd > gpurun_out/ic2.log 2>&1
python - <<'PY'
import csv, collections
for f in ("gpurun_out/ic/ic_counter_collection.csv", "gpurun_out/ic2/ic2_counter_collection.csv"):
    try:
        rows = list(csv.DictReader(open(f)))
    except Exception as e:
        print(f, e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "conv_px" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: (sum(x) / len(x), len(x)) for c, x in v.items()})
PY
tail -3 gpurun_out/ic.log
