#!/bin/bash
# stage timings of the 4-byte-record voxel path for every partition geometry (one process per geometry)
mkdir -p gpurun_out; out=gpurun_out/v3_sweep.txt; : > $out
EVK_V3_EPT=16 timeout 600 python tools/v3_sweep.py --check --v2 --big --scenes >> $out 2>&1
for e in ${EPTS:-12 8}; do EVK_V3_EPT=$e timeout 300 python tools/v3_sweep.py --big >> $out 2>&1; done
grep -v amdgpu.ids $out
