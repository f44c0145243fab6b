#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/v2_sweep.txt
: > $out
for part in 512x32 1024x8s 1024x16; do
  EVK_V2_PART=$part timeout 300 python tools/v2_sweep.py --big --v2only ${CHECK} >> $out 2>&1
done
for wg in 256 1024; do
  EVK_V2_PART=1024x8s EVK_V2_WG=$wg timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
done
EVK_V2_PART=1024x8s EVK_V2_U=4 timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
grep -v amdgpu.ids $out
