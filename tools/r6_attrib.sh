#!/bin/bash
# round 6, VERDICT item 1: where the 720p voxel tile kernel's time goes.  (a) sensor x record size x event count table,
# (b) the 50 M / 720p / 4-byte-record case under the tile kernel's ablation builds and load / entry geometry variants.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
{
python tools/tile_attrib.py
for v in product ablB1 ablB2 ablB3 u1 u3 u4 ent1 ent2 ent4 product; do
  lib=$PWD/tools/exp/libevk_$v.so; [ $v = product ] && lib=$PWD/event_utils_amd/csrc/libevk.so
  echo "== $v"; EVK_LIB_PATH=$lib timeout 300 python tools/tile_attrib.py --case 720x1280x50000000x4 2>&1 | grep -v "^lib"
done
} > gpurun_out/r6_attrib.txt 2>&1
tail -50 gpurun_out/r6_attrib.txt
