#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
{
for t in 38x24 64x15 32x30 40x24 32x24 20x48 80x12 128x7 38x24; do
  echo "== tile $t"; timeout 300 python tools/tile_attrib.py --tile $t --case 720x1280x50000000x4 2>&1 | grep -v "^lib\|amdgpu.ids"
done
} > gpurun_out/r6_shapes.txt 2>&1
cat gpurun_out/r6_shapes.txt
