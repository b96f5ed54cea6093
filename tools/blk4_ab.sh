for cfg in "morphomnist" "cmnist --dmol" "mimic224"; do
  for on in 0 1; do
    echo "== $cfg CGEN_BLK4=$on"
    CGEN_BLK4=$on python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu --no-cf --no-f32 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['launches_per_step'], d['elbo_nats_per_dim'])"
  done
done
