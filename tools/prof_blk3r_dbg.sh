#!/bin/bash
export TMPDIR=/tmp
for r in $1; do for d in $2; do rm -rf gpurun_out/pb; CGEN_BLK3R_DBG=$d CGEN_CONV_TRACE=1 timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/pb -o pb --output-format csv -- python tools/bench_blk3r.py $r 20 > gpurun_out/pb.log 2>&1 || echo "key $r dbg $d: FAILED / timed out"; grep blk3r gpurun_out/pb.log | sort | uniq | head -2 | cut -c1-140; python - <<PY
import csv,glob
fs=glob.glob("gpurun_out/pb/**/*kernel_stats.csv",recursive=True)
for r in (csv.DictReader(open(fs[0])) if fs else []):
    if "blk3r" in r["Name"]: print("key $r dbg $d: %-44s avg %.2f us" % (r["Name"][:44], float(r["AverageNs"])/1e3))
PY
done; done; rm -rf gpurun_out/pb gpurun_out/pb.log
