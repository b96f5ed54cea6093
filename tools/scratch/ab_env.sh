# usage: ab_env.sh VAR v1 v2 [reps]  -- interleaved A/B of bench.py under one environment variable
run() { env $1=$2 python bench.py --no-cpu --no-f32 --no-extra --no-cf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$2', d['value'], d['ms_per_step'], d['launches_per_step'])"; }
for i in $(seq ${4:-2}); do run $1 $2; run $1 $3; done
