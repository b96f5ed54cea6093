#!/bin/bash
# kernel-trace timing of a wide-resolution light Block: row-streaming instance (FUSE=2) vs two launches (FUSE=0)
# usage: tools/prof_blk3r.sh "<keys: 192 1922 96 962>"
export TMPDIR=/tmp
for r in $1; do for f in 2 0; do rm -rf gpurun_out/pb; FUSE=$f timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/pb -o pb --output-format csv -- python tools/bench_blk3r.py $r 20 > /dev/null 2>&1 || echo "key $r fuse $f: FAILED / timed out"; python - <<PY
import csv,glob
fs=glob.glob("gpurun_out/pb/**/*kernel_stats.csv",recursive=True)
rows=[r for r in (csv.DictReader(open(fs[0])) if fs else []) if any(k in r["Name"] for k in ("blk3","conv_px","conv_ws","conv_tile","conv_kernel"))]
for r in rows: print("key $r fuse $f: %-60s calls %3s avg %.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done; done; rm -rf gpurun_out/pb
