#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/v2_sweep.txt
: > $out
EVK_V2_WG=512 timeout 300 python tools/v2_sweep.py --big --v2only --check >> $out 2>&1
for wg in 256 1024; do
  EVK_V2_WG=$wg timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
done
EVK_V2_WG=512 EVK_V2_U=4 timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
EVK_V2_WG=1024 EVK_V2_U=4 timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
grep -v amdgpu.ids $out
