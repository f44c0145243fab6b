#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/v2_sweep.txt
: > $out
export EVK_LIB_PATH=$PWD/tools/ablate/libevk_lb11.so EVK_V2_LB=11
for cfg in "6x5 512" "6x5 1024"; do
  set -- $cfg
  echo "TILE=$1 WG=$2" >> $out
  EVK_VOXEL_TILE=$1 EVK_V2_WG=$2 timeout 120 python tools/v2_sweep.py --big --v2only >> $out 2>&1
done
grep -v amdgpu.ids $out
