#!/bin/bash
# Round evidence on the GPU box (writes under gpurun_out/<tag>*): kernel stats of the default bench command, the two PMC traffic
# passes, the instruction-mix passes, the default bench line with the shape table, and one bench line per other BASELINE config.
# usage: tools/collect_round.sh <tag>
tag=$1
export TMPDIR=/tmp
cd /root/repo
rm -rf gpurun_out/$tag gpurun_out/${tag}_f gpurun_out/${tag}_w gpurun_out/${tag}_A gpurun_out/${tag}_B
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o $tag --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu --no-f32 --no-extra > gpurun_out/$tag.log 2>&1
python tools/stats_to_txt.py gpurun_out/$tag/*/${tag}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu --no-f32   (ukbb192, batch 32, f16, 1 x MI355X; 27 train steps incl. 20 preparation steps + the eager profiling step, then the counterfactual loop)" > gpurun_out/${tag}_kernel_stats.txt 2>/dev/null || python tools/stats_to_txt.py $(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu --no-f32 (ukbb192, batch 32, f16, 1 x MI355X)" > gpurun_out/${tag}_kernel_stats.txt
python tools/timeline.py $(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1) gpurun_out/${tag}_step_timeline.txt > /dev/null 2>&1
rm -rf gpurun_out/$tag/*/*kernel_trace.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_f -o f --output-format csv -- python bench.py --steps 2 --warmup 1 --prep-steps 1 --no-cpu --no-cf --no-f32 --no-extra > gpurun_out/${tag}_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_w -o w --output-format csv -- python bench.py --steps 2 --warmup 1 --prep-steps 1 --no-cpu --no-cf --no-f32 --no-extra > gpurun_out/${tag}_w.log 2>&1
python tools/pmc_traffic.py $(find gpurun_out/${tag}_f -name "*counter_collection.csv" | head -1) $(find gpurun_out/${tag}_w -name "*counter_collection.csv" | head -1) gpurun_out/${tag}_traffic.json > /dev/null
bash tools/pmc_insts.sh ${tag} python bench.py --steps 1 --warmup 1 --prep-steps 1 --no-cpu --no-cf --no-f32 --no-extra
python tools/pmc_insts.py gpurun_out/${tag} > gpurun_out/${tag}_instruction_mix.txt 2>&1
python tools/pmc_classes.py gpurun_out/${tag} > gpurun_out/${tag}_mfma_busy.json 2>&1
rm -rf gpurun_out/${tag}_f gpurun_out/${tag}_w gpurun_out/${tag}_A gpurun_out/${tag}_B gpurun_out/$tag
CGEN_SHAPE_DUMP=gpurun_out/${tag}_shapes.txt timeout 400 python bench.py > gpurun_out/${tag}_bench_ukbb192.json 2> gpurun_out/${tag}_bench_ukbb192.err
# (the other BASELINE configs are inside the default line since round 3: "configs")
CGEN_STAGE=1 timeout 120 python tools/stage_stamps.py ukbb192 32 4 > gpurun_out/${tag}_stage_stamps_ukbb192.txt 2>/dev/null
CGEN_STAGE=1 timeout 120 python tools/stage_stamps.py morphomnist 256 3 > gpurun_out/${tag}_stage_stamps_morphomnist.txt 2>/dev/null
for st in 0 1; do CGEN_STAGE=$st timeout 200 python bench.py --no-cpu --no-f32 --no-extra > gpurun_out/${tag}_bench_ukbb192_stage$st.json 2>/dev/null; CGEN_STAGE=$st timeout 200 python bench.py --no-cpu --no-f32 --no-extra --config morphomnist > gpurun_out/${tag}_bench_morphomnist_stage$st.json 2>/dev/null; done
ls -la gpurun_out | grep $tag
