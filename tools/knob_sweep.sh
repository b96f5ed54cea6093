#!/bin/bash
# Interleaved one-knob sweeps of the default bench workload (ukbb192 B=32 f16): prints images/s per setting, twice.
# usage: tools/knob_sweep.sh   (on the GPU box; ~40 s per run)
cd /root/repo
run() {  # label, env assignments...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --no-cpu --no-extra --no-f32 --no-cf 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['value'],1), round(d['ms_per_step'],3))"
}
for rep in 1 2; do
  run base X=1
  run minwg384 CGEN_PX_MINWG=384
  run minwg768 CGEN_PX_MINWG=768
  run maxnp1 CGEN_PX_MAXNP=1
  run maxnp3 CGEN_PX_MAXNP=3
  run smallp4000 CGEN_SMALLP_MAXP=4000
  run smallp9000 CGEN_SMALLP_MAXP=9000
  run bgwgs256 CGEN_WGRAD_BG_WGS=256
  run bgwgs352 CGEN_WGRAD_BG_WGS=352
  run flush52 CGEN_WGRAD_FLUSH_FRAC=0.52
  run flush64 CGEN_WGRAD_FLUSH_FRAC=0.64
  run pxlds40 CGEN_PX_LDS=40
done
