#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
{
for v in product w6e1 e2 w5 u3 u1 product; do
  lib=$PWD/tools/exp/libevk_$v.so; [ $v = product ] && lib=$PWD/event_utils_amd/csrc/libevk.so
  echo "== $v"; EVK_LIB_PATH=$lib timeout 300 python tools/tile_attrib.py --case 720x1280x50000000x4 2>&1 | grep -v "^lib\|amdgpu.ids"
done
} > gpurun_out/r6_occ.txt 2>&1
cat gpurun_out/r6_occ.txt
