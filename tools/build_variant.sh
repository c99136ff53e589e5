#!/bin/bash
# A/B measurement helper (not part of the product): builds libsimx_hip.so with extra -D flags on gemm.hip into
# tools/variants/<name>/ and a kbench linked against it.   tools/build_variant.sh B -DSIMX_P3_HALF
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=tools/variants/$name
mkdir -p $d
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
hipcc $FLAGS "$@" -c simxns_amd/csrc/gemm.hip -o $d/gemm.o &
hipcc $FLAGS "$@" -c simxns_amd/csrc/gemm_p5.hip -o $d/gemm_p5.o &
hipcc $FLAGS "$@" -c simxns_amd/csrc/gemm_tn5.hip -o $d/gemm_tn5.o &
wait
OBJS="$d/gemm.o $d/gemm_p5.o $d/gemm_tn5.o"
for f in gemm_x3 gemm_xp attention attention_f32 attention_x3 layernorm loss sampler optim encoder collate retrieval det; do OBJS="$OBJS simxns_amd/csrc/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $d/libsimx_hip.so
hipcc -O2 tools/kbench.cpp -Iinclude -L$d -lsimx_hip -ldl -Wl,-rpath,'$ORIGIN' -o $d/kbench
echo "built $d"
