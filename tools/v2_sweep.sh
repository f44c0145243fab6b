#!/bin/bash
# One tools/v2_sweep.py process per kernel-geometry variant of the one-pass voxel path (the variables are read when the
# library loads).  PARTS / WGS / US override the lists; LIBS = ablation libraries under tools/ablate/ to time as well.
mkdir -p gpurun_out
out=gpurun_out/v2_sweep.txt
: > $out
timeout 300 python tools/v2_sweep.py --check --big >> $out 2>&1 || { cat $out; exit 1; }
for part in ${PARTS:-1024x8s 1024x12s 1024x16 512x32}; do
  EVK_V2_PART=$part timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
done
for wg in ${WGS:-256 1024}; do for u in ${US:-2 4}; do
  EVK_V2_WG=$wg EVK_V2_U=$u timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
done; done
for l in ${LIBS}; do
  EVK_LIB_PATH=$PWD/tools/ablate/$l timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
done
grep -v amdgpu.ids $out
