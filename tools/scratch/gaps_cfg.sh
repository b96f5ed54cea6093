export TMPDIR=/tmp
rm -rf gpurun_out/gp
rocprofv3 --kernel-trace -d gpurun_out/gp -o gp --output-format csv -- python bench.py --config $1 $2 --steps 5 --warmup 2 --no-cpu --no-f32 --no-extra --no-cf > gpurun_out/gp.log 2>&1
python tools/step_gaps.py $(find gpurun_out/gp -name "*kernel_trace.csv" | head -1) | head -24
python tools/scratch/qcount.py $(find gpurun_out/gp -name "*kernel_trace.csv" | head -1)
python tools/step_breakdown.py $(find gpurun_out/gp -name "*kernel_trace.csv" | head -1) | head -16
rm -rf gpurun_out/gp
