#!/bin/bash
# voxel tests + stage timings of the default one-pass path
mkdir -p gpurun_out; out=gpurun_out/v2_run.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_native.py -x -q -m gpu -k "voxel or native" 2>&1 | tail -4 >> $out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -m gpu -k "f3 or f16 or f15 or c2 or voxel" 2>&1 | tail -4 >> $out
timeout 300 python tools/voxel_sweep.py --scenes --big --native 2>&1 | grep "^v\|^native"  >> $out
for g in ${GEOS:-1024x12 1024x16 512x16}; do
  echo "== EVK_V2_PART=$g" >> $out
  EVK_LIB_PATH=$PWD/tools/exp/libevk_exp.so EVK_V2_PART=$g timeout 300 python tools/voxel_sweep.py --big 2>&1 | grep "^v"  >> $out
done
cat $out
