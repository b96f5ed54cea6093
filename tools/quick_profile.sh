#!/bin/bash
# Quick perf loop on the GPU box: bench line (ukbb192 B=32 f16, no extras) + kernel stats + step timeline under gpurun_out/<tag>_*.
# usage: tools/quick_profile.sh <tag> [extra bench args / env via env]
tag=$1; shift
export TMPDIR=/tmp
cd /root/repo
python bench.py --steps 20 --warmup 5 --no-cpu --no-f32 --no-extra --no-cf "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rm -rf gpurun_out/$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o $tag --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu --no-f32 --no-extra --no-cf "$@" > gpurun_out/$tag.log 2>&1
python tools/stats_to_txt.py $(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1) "quick profile $tag; code tree $(python tools/tree_sha.py)" > gpurun_out/${tag}_kernel_stats.txt
python tools/timeline.py $(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1) gpurun_out/${tag}_step_timeline.txt > /dev/null 2>&1
rm -rf gpurun_out/$tag
python - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
print("BENCH", d["value"], d["unit"], d["ms_per_step"], "ms/step", "roofline", d.get("roofline", {}).get("frac"), "launches", d.get("launches_per_step"))
PY
head -32 gpurun_out/${tag}_step_timeline.txt
