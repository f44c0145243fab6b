#!/bin/bash
# 4-byte against 8-byte records of the one-pass voxel path: voxel tests under both, then alternating stage timings
mkdir -p gpurun_out; out=gpurun_out/rec_ab.txt; : > $out
for r in 4 8; do
  echo "== tests EVK_V2_REC=$r" >> $out
  EVK_V2_REC=$r timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_native.py tests/test_gpu_parity.py -x -q -m gpu -k "voxel or native or f3 or f15 or f16" 2>&1 | grep -E "passed|failed|Error" | tail -3 >> $out
done
EVK_V2_REC=4 timeout 300 python tools/voxel_sweep.py --check 2>&1 | grep "^check\|^clustered\|^determ\|Error" >> $out
for i in 1 2; do for r in 4 8; do
  echo "== EVK_V2_REC=$r" >> $out
  EVK_V2_REC=$r timeout 300 python tools/voxel_sweep.py --scenes --big --native 2>&1 | grep "^v\|^native"  | cut -c1-185 >> $out
done; done
cat $out
