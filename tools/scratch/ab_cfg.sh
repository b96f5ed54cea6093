# usage: ab_cfg.sh CONFIG VAR v1 v2 [reps] [extra bench args]
run() { env $2=$3 python bench.py --config $1 $5 --no-cpu --no-f32 --no-extra --no-cf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2=$3', d['value'], d['ms_per_step'], d['launches_per_step'])"; }
for i in $(seq ${5:-2}); do run $1 $2 $3 x "$6"; run $1 $2 $4 x "$6"; done
