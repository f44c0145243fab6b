#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/v2_sweep.txt
: > $out
for st in 0 1 2 3 4 6; do
    echo "STAGGER=$st" >> $out
    EVK_V2_STAGGER=$st timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
done
grep -v amdgpu.ids $out
