#!/bin/bash
# Build libsimx_hip.so for gfx950 (in-tree; the .so travels with the snapshot to the GPU box).
# Every compile also records the compiler's per-kernel resource report (registers, scratch, spills, occupancy) in <unit>.res;
# tools/check_isa.py reads those and fails the build when a kernel outside its allow-list spills or uses scratch.
set -e
cd "$(dirname "$0")"
OUT=../libsimx_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -Rpass-analysis=kernel-resource-usage"
OBJS=""
PIDS=""
for f in gemm gemm_p5 gemm_tn5 gemm_x3 gemm_xp attention attention_f32 attention_x3 layernorm loss sampler optim encoder collate retrieval det; do
  if [ ! -f $f.o ] || [ ! -f $f.res ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ prof.h -nt $f.o ] || [ p3.h -nt $f.o ] || [ ../../include/simx.h -nt $f.o ]; then
    echo "hipcc $f.hip"
    ( hipcc $FLAGS -c $f.hip -o $f.o 2> $f.res.tmp || { grep -v "kernel-resource-usage" $f.res.tmp | head -60 >&2; rm -f $f.o; exit 1; }
      grep -E "remark: +(Function Name|ScratchSize|VGPRs|AGPRs|Occupancy|LDS Size)" $f.res.tmp | sed -E 's/^[^ ]+ remark: +//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' > $f.res
      grep -vE "kernel-resource-usage|^ +[0-9]* *\||^ +\| *\^|^[0-9]+ warnings? generated" $f.res.tmp >&2 || true
      rm -f $f.res.tmp ) &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $f.o"
done
for p in $PIDS; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
echo "built $OUT"
