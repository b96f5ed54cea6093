for v in 0.58 0.5 0.66 0.74 0.45,0.8 0.58,0.88 0.58; do
echo "FLUSH_FRAC=$v: $(CGEN_WGRAD_FLUSH_FRAC=$v python bench.py --no-cpu --no-f32 --no-extra --no-cf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
