#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/v2_sweep.txt; : > $out
timeout 300 python tools/v2_sweep.py --big --v2only --check >> $out 2>&1
for u in ${US:-4}; do EVK_V2_U=$u timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1; done
grep -v amdgpu.ids $out
