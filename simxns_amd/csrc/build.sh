#!/bin/bash
# Build libsimx_hip.so for gfx950 (in-tree; the .so travels with the snapshot to the GPU box).
set -e
cd "$(dirname "$0")"
OUT=../libsimx_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
OBJS=""
for f in gemm gemm_x3 gemm_xp attention attention_f32 attention_x3 layernorm loss sampler optim encoder collate retrieval det; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ prof.h -nt $f.o ] || [ p3.h -nt $f.o ] || [ ../../include/simx.h -nt $f.o ]; then
    echo "hipcc $f.hip"
    hipcc $FLAGS -c $f.hip -o $f.o &
  fi
  OBJS="$OBJS $f.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
echo "built $OUT"
