#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/v2_sweep.txt
: > $out
for cfg in "512 4" "256 2" "1024 2" "1024 4" "256 4"; do
  set -- $cfg
  EVK_V2_WG=$1 EVK_V2_U=$2 timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
done
grep -v amdgpu.ids $out
