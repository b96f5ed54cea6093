#!/bin/bash
# Round-end evidence on the GPU box: kernel stats, the two PMC traffic passes, the default bench line.  usage: tools/collect_profiles.sh <tag>
tag=$1
export TMPDIR=/tmp
cd /root/repo
rm -rf gpurun_out/$tag gpurun_out/${tag}_f gpurun_out/${tag}_w
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o $tag --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/$tag.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_f -o f --output-format csv -- python bench.py --steps 2 --warmup 1 --prep-steps 1 --no-cpu --no-cf > gpurun_out/${tag}_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_w -o w --output-format csv -- python bench.py --steps 2 --warmup 1 --prep-steps 1 --no-cpu --no-cf > gpurun_out/${tag}_w.log 2>&1
python tools/pmc_traffic.py gpurun_out/${tag}_f/f_counter_collection.csv gpurun_out/${tag}_w/w_counter_collection.csv gpurun_out/${tag}_traffic.json
python bench.py > gpurun_out/${tag}_bench.log 2>&1
tail -1 gpurun_out/${tag}_bench.log
