#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/v2_sweep.txt
: > $out
timeout 120 python tools/v2_sweep.py --big --v2only --check >> $out 2>&1
grep -v amdgpu.ids $out
