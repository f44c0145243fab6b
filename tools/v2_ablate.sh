#!/bin/bash
# Stage ablation of the one-pass voxel kernels (libraries from tools/v2_ablate_build.sh) + tile-kernel geometry variants.
mkdir -p gpurun_out
out=gpurun_out/v2_ablate.txt
: > $out
export EVK_V2_WG=${EVK_V2_WG:-512} EVK_V2_G=${EVK_V2_G:-4}
for l in a0 a1 a2 a3 b0 b1 b2; do
  EVK_LIB_PATH=$PWD/tools/ablate/libevk_$l.so timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
done
for cfg in "512 4" "1024 4" "1024 8"; do
  set -- $cfg
  EVK_V2_WG=$1 EVK_V2_G=$2 timeout 300 python tools/v2_sweep.py --big --v2only >> $out 2>&1
done
grep -v "amdgpu.ids" $out
