#!/bin/bash
# voxel tests + stage timings of the default one-pass path (balanced tiles) against the old power-of-two tiling
mkdir -p gpurun_out; out=gpurun_out/v2_run.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_native.py -x -q -m gpu -k "voxel or native" 2>&1 | tail -15 >> $out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f3 or f16 or f15 or c2" 2>&1 | tail -5 >> $out
EVK_VOXEL_PATH=v2 timeout 300 python tools/v3_sweep.py --scenes --big 2>&1 | grep "^v" | sed 's/^v3/v2 (balanced)/' >> $out
EVK_VOXEL_PATH=v2 EVK_VOXEL2_TILE=32x16 timeout 300 python tools/v3_sweep.py --scenes 2>&1 | grep "^v" | sed 's/^v3/v2 (32x16)/' >> $out
cat $out
