#!/bin/bash
# HBM traffic of the default Block at 224x224 / 112x112 / 28x28 (batch 32), fused (cgen_block4) against four launches: FETCH_SIZE and
# WRITE_SIZE in separate passes (MI355X guide), per mode.  usage (GPU box): bash tools/pmc_blk4_traffic.sh <tag>
tag=$1
export TMPDIR=/tmp
for mode in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    B4_ONLY=$mode rocprofv3 --kernel-trace --pmc $c -d gpurun_out/${tag}_$c$mode -o t --output-format csv -- python tools/bench_blk4.py 4 0,1,5 > /dev/null 2>&1
  done
  python - $tag $mode <<'PY'
import csv, glob, sys, collections
tag, mode = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for c, k in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
    f = glob.glob(f"gpurun_out/{tag}_{c}{mode}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        n = r["Kernel_Name"]
        if not any(s in n for s in ("blk4", "conv_px", "conv_ws", "conv_tile", "conv_smallp", "conv_kernel")):
            continue
        key = (n.split("(")[0].replace("void cgen::", ""), r["Grid_Size"])
        tot[key][k] += float(r["Counter_Value"])
        if k == 0:
            tot[key][2] += 1
print("mode", "fused" if mode == "1" else "four launches", "(HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) KB, gfx950 correction of the guide)")
for key, (f, w, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    if n:
        print("  %-44s grid %-9s x%-3d  %8.1f MB per launch (fetch %7.1f, write %7.1f)" % (key[0][:44], key[1], n, (2 * f + w) / 1024 / n, 2 * f / 1024 / n, w / 1024 / n))
PY
  rm -rf gpurun_out/${tag}_FETCH_SIZE$mode gpurun_out/${tag}_WRITE_SIZE$mode
done
