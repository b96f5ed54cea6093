#!/bin/bash
# kernel-trace timing of the small-image fused Block under ablation bits (CGEN_BLK3S_DBG); every process under its own timeout
# usage: tools/prof_blk3s.sh "<res list>" "<dbg list>"
export TMPDIR=/tmp
for r in $1; do for d in $2; do rm -rf gpurun_out/pb; CGEN_BLK3S_DBG=$d timeout 90 rocprofv3 --kernel-trace --stats -d gpurun_out/pb -o pb --output-format csv -- python tools/bench_blk3s.py $r 50 > /dev/null 2>&1 || echo "res $r dbg $d: FAILED / timed out"; python - <<PY
import csv,glob
fs=glob.glob("gpurun_out/pb/**/*kernel_stats.csv",recursive=True)
for r in (csv.DictReader(open(fs[0])) if fs else []):
    if "blk3s" in r["Name"]: print("res $r dbg $d: avg %.2f us min %.2f calls %s" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, r["Calls"]))
PY
done; done; rm -rf gpurun_out/pb
