#!/bin/bash
# Build libcgen_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=libcgen_hip.so
SRCS="csrc/runtime.hip csrc/conv.hip csrc/elementwise.hip csrc/latent.hip csrc/likelihood.hip csrc/optim.hip"
mkdir -p build
OBJS=""
pids=""
for s in $SRCS; do
  o=build/$(basename ${s%.hip}).o
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ csrc/common.h -nt "$o" ] || [ ../include/cgen_hip.h -nt "$o" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c "$s" -o "$o" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo "built $(pwd)/$OUT"
