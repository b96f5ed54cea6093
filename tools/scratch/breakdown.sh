export TMPDIR=/tmp
rm -rf gpurun_out/bd
rocprofv3 --kernel-trace -d gpurun_out/bd -o bd --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu --no-f32 --no-extra --no-cf > gpurun_out/bd.log 2>&1
python tools/step_breakdown.py $(find gpurun_out/bd -name "*kernel_trace.csv" | head -1)
rm -rf gpurun_out/bd
