#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
{
for i in 1 2; do
for v in product nopf; do
  lib=$PWD/tools/exp/libevk_$v.so; [ $v = product ] && lib=$PWD/event_utils_amd/csrc/libevk.so
  echo "== $v"; EVK_LIB_PATH=$lib timeout 300 python tools/tile_attrib.py --case 720x1280x50000000x4 2>&1 | grep -v "^lib\|amdgpu.ids"
  EVK_LIB_PATH=$lib timeout 300 python tools/tile_attrib.py --case 480x640x10000000 2>&1 | grep -v "^lib\|amdgpu.ids"
done; done
EVK_LIB_PATH=$PWD/tools/exp/libevk_phase.so python tools/tile_phases.py 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_parity.py -x -q -m gpu -k "voxel" 2>&1 | tail -2
} > gpurun_out/r6_fast.txt 2>&1
cat gpurun_out/r6_fast.txt
