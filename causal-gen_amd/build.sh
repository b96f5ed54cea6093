#!/bin/bash
# Build libcgen_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# NOPK: no packed-f32 VALU instructions (v_pk_mul/add/fma_f32) in any kernel.  Measured on MI355X (tools/coexec_probe.py):
# a wave executing them returns corrupted results when a wave of ANOTHER kernel on the same SIMD is executing
# v_mfma_f32_16x16x32_bf16 -- exactly what the background weight-gradient kernel does next to the backward chain.
# (The feature name is unknown to the host pass, which says so and ignores it.)
set -e
cd "$(dirname "$0")"
OUT=libcgen_hip.so
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
SRCS="csrc/runtime.hip csrc/block.hip csrc/block4.hip csrc/conv.hip csrc/wgrad3.hip csrc/elementwise.hip csrc/latent.hip csrc/likelihood.hip csrc/optim.hip"
mkdir -p build
OBJS=""
pids=""
for s in $SRCS; do
  o=build/$(basename ${s%.hip}).o
  OBJS="$OBJS $o"
  stale=0
  for dep in "$s" csrc/common.h csrc/wgrad3.h csrc/*.inc ../include/cgen_hip.h; do
    if [ "$dep" -nt "$o" ]; then stale=1; fi
  done
  if [ ! -f "$o" ] || [ $stale = 1 ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $NOPK ${CGEN_EXTRA_FLAGS} -Wno-unused-result -c "$s" -o "$o" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo "built $(pwd)/$OUT"
