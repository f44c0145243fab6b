#!/bin/bash
# run after tools/v3_ablate.sh: stage timings of every ablation build (10 M events VGA, 50 M events 720p)
mkdir -p gpurun_out; out=gpurun_out/v3_ablate.txt; : > $out
for v in P1 P2 P3 P4 T1 T2 T3 T4; do
  echo "== $v" >> $out
  EVK_LIB_PATH=$PWD/tools/ablate/libevk_$v.so timeout 300 python tools/v3_sweep.py --big 2>&1 | grep "^v3" >> $out
done
echo "== full" >> $out
timeout 300 python tools/v3_sweep.py --big 2>&1 | grep "^v3" >> $out
cat $out
