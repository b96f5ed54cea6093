export TMPDIR=/tmp
for t in 8 0; do
rm -rf gpurun_out/ks$t
CGEN_BLK3_TH=$t rocprofv3 --kernel-trace --stats -d gpurun_out/ks$t -o ks --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu --no-f32 --no-extra --no-cf > gpurun_out/ks$t.log 2>&1
echo "== TH env $t"; python tools/stats_to_txt.py $(find gpurun_out/ks$t -name "*kernel_stats.csv" | head -1) x | grep "blk3\|mega" | cut -c1-150
rm -rf gpurun_out/ks$t
done
