run() { echo "$1 RES=$2 RES3=$3: $(CGEN_BLK3_RES=$2 CGEN_BLK3_RES3=$3 python bench.py --config $1 --no-cpu --no-f32 --no-extra --no-cf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['launches_per_step'])")"; }
run mimic224 1 1
run mimic224 20-64 20-112
run mimic224 20-64 20-64
run mimic224 20-112 20-112
run ukbb192 20-64 20-112
run morphomnist 20-64 20-112
run morphomnist 16-64 16-112
