run() { env $1=$2 python bench.py --no-cpu --no-f32 --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$2', d['value'], d['counterfactuals_per_s'], d.get('cf_plain_trunk'))"; }
for i in 1 2; do run $1 $2; run $1 $3; done
