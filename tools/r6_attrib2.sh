#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
{
for v in product ablB1 ld1 ld2 ld1B1 ld2B1 product; do
  lib=$PWD/tools/exp/libevk_$v.so; [ $v = product ] && lib=$PWD/event_utils_amd/csrc/libevk.so
  echo "== $v"; EVK_LIB_PATH=$lib timeout 300 python tools/tile_attrib.py --case 720x1280x50000000x4 2>&1 | grep -v "^lib\|amdgpu.ids"
done
} > gpurun_out/r6_attrib2.txt 2>&1
cat gpurun_out/r6_attrib2.txt
