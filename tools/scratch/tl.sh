export TMPDIR=/tmp
rm -rf gpurun_out/gp
rocprofv3 --kernel-trace -d gpurun_out/gp -o gp --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu --no-f32 --no-extra --no-cf > gpurun_out/gp.log 2>&1
python tools/timeline.py $(find gpurun_out/gp -name "*kernel_trace.csv" | head -1) /dev/stdout | head -26
python tools/scratch/qcount.py $(find gpurun_out/gp -name "*kernel_trace.csv" | head -1)
python tools/step_gaps.py $(find gpurun_out/gp -name "*kernel_trace.csv" | head -1) | tail -12
rm -rf gpurun_out/gp
