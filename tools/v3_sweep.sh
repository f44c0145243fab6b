#!/bin/bash
# stage timings of the 4-byte-record voxel path for every partition geometry (one process per geometry; needs
# tools/exp_build.sh first)
mkdir -p gpurun_out; out=gpurun_out/v3_sweep.txt; : > $out
export EVK_LIB_PATH=$PWD/tools/exp/libevk_exp.so
EVK_V3_GEO=1024x16x0 timeout 600 python tools/v3_sweep.py --check --v2 --big --scenes >> $out 2>&1
for g in ${GEOS:-1024x12x0 1024x8x0 1024x8x1 1024x12x1 512x16x0 512x16x1 512x12x1 512x8x1}; do
  EVK_V3_GEO=$g timeout 300 python tools/v3_sweep.py --big >> $out 2>&1
done
grep -v amdgpu.ids $out
