#!/bin/bash
# hot-tile split rule of the one-pass voxel path: scenes x EVK_V2_SPLIT="at,part" (x EVK_V2_FORCE_RESIDENCY)
mkdir -p gpurun_out; out=gpurun_out/v2_split.txt; : > $out
for f in ${FS:-4,4 3,1 3,1.5 2.5,1 4,1 3,0.75}; do
  for r in ${RS:-1}; do
  echo "== EVK_V2_SPLIT=$f EVK_V2_FORCE_RESIDENCY=$r" >> $out
  EVK_V2_FORCE_RESIDENCY=$r EVK_V2_SPLIT=$f timeout 300 python tools/voxel_sweep.py --scenes ${BIG:+--big} 2>&1 | grep "^v"  >> $out
  done
done
cat $out
