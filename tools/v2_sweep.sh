#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/v2_sweep.txt
: > $out
for l in "" ${LIBS}; do
  for part in ${PARTS:-512x32 1024x16}; do
    lp=""; [ -n "$l" ] && lp=$PWD/tools/ablate/$l
    EVK_LIB_PATH=$lp EVK_V2_PART=$part timeout 300 python tools/v2_sweep.py --big --v2only ${CHECK} >> $out 2>&1
  done
done
grep -v amdgpu.ids $out
