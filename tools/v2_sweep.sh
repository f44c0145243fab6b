#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/v2_sweep.txt
: > $out
for wg in 256 512 1024; do for u in 2 4 8; do
  EVK_V2_WG=$wg EVK_V2_U=$u timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
done; done
grep -v amdgpu.ids $out
