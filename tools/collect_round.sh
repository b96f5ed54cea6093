#!/bin/bash
# Round evidence on the GPU box (writes under gpurun_out/<tag>*): kernel stats + step timeline of the default bench command, the PMC
# traffic passes (f16 and f32 legs), the instruction-mix / MFMA-busy passes, the default bench line with the shape table.  Every JSON
# summary carries _source.code_tree_sha (tools/tree_sha.py) of the code it was taken on; bench.py prints `stale` against the running tree
# and tools/check_profiles.py refuses summaries of another tree.
# usage: tools/collect_round.sh <tag>
tag=$1
export TMPDIR=/tmp
cd /root/repo
rm -rf gpurun_out/$tag gpurun_out/${tag}_f gpurun_out/${tag}_w gpurun_out/${tag}_A gpurun_out/${tag}_B
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o $tag --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu --no-f32 --no-extra > gpurun_out/$tag.log 2>&1
python tools/stats_to_txt.py $(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu --no-f32 --no-extra  (ukbb192, batch 32, f16, 1 x MI355X; 27 train steps incl. 20 preparation steps + the eager profiling step, then the counterfactual loop); code tree $(python tools/tree_sha.py)" > gpurun_out/${tag}_kernel_stats.txt
python tools/timeline.py $(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1) gpurun_out/${tag}_step_timeline.txt > /dev/null 2>&1
rm -rf gpurun_out/$tag
for dt in f16 f32; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_f -o f --output-format csv -- python bench.py --dtype $dt --steps 2 --warmup 1 --prep-steps 1 --no-cpu --no-cf --no-f32 --no-extra > gpurun_out/${tag}_f.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_w -o w --output-format csv -- python bench.py --dtype $dt --steps 2 --warmup 1 --prep-steps 1 --no-cpu --no-cf --no-f32 --no-extra > gpurun_out/${tag}_w.log 2>&1
  python tools/pmc_traffic.py $(find gpurun_out/${tag}_f -name "*counter_collection.csv" | head -1) $(find gpurun_out/${tag}_w -name "*counter_collection.csv" | head -1) gpurun_out/${tag}_hbm_traffic_${dt}_b32.json > /dev/null
  rm -rf gpurun_out/${tag}_f gpurun_out/${tag}_w
done
bash tools/pmc_insts.sh ${tag} python bench.py --steps 1 --warmup 1 --prep-steps 1 --no-cpu --no-cf --no-f32 --no-extra
python tools/pmc_insts.py gpurun_out/${tag} > gpurun_out/${tag}_instruction_mix.txt 2>&1
python tools/pmc_classes.py gpurun_out/${tag} > gpurun_out/${tag}_mfma_busy_by_class.json 2>gpurun_out/${tag}_mfma_busy.err
rm -rf gpurun_out/${tag}_A gpurun_out/${tag}_B
ls -la gpurun_out | grep $tag
