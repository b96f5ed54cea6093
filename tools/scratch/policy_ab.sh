run() { echo "RES=$1 RES3=$2 NOSM=$3: $(CGEN_BLK3_RES=$1 CGEN_BLK3_RES3=$2 CGEN_BLK3_NOSM=$3 python bench.py --no-cpu --no-f32 --no-extra --no-cf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['launches_per_step'])")"; }
run 24,48 24,48,96 0
run 24,48 24,48 0
run 24,48,96 24,48,96 1
run 24,48,96 24,48,96 0
run 24,48,96,192 24,48,96 0
run 24,48 24,48,96 0
