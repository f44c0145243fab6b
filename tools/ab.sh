#!/bin/bash
# A/B of builds of libevk.so on the SAME box (boxes differ by ~10 %): alternating runs of the stage timings
# usage: tools/ab.sh name1 name2 ...   (tools/exp/libevk_<name>.so; "default" = the product library)
for i in 1 2; do
  for name in "$@"; do
    lib=tools/exp/libevk_$name.so; [ "$name" = default ] && lib=event_utils_amd/csrc/libevk.so
    echo "== $name"; EVK_LIB_PATH=$PWD/$lib python tools/voxel_sweep.py ${BIG:+--big} ${ROTATE:+--rotate} 2>&1 | grep "^v\|Error\|error"  | cut -c1-190
  done
done
