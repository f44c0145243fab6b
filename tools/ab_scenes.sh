#!/bin/bash
# as tools/ab.sh, with the structured scenes (moving edges, blob) at 10 M events
for i in 1 2; do
  for name in "$@"; do
    lib=tools/exp/libevk_$name.so; [ "$name" = default ] && lib=event_utils_amd/csrc/libevk.so
    echo "== $name"; EVK_LIB_PATH=$PWD/$lib python tools/voxel_sweep.py --scenes ${BIG:+--big} 2>&1 | grep "^v\|Error\|error"  | cut -c1-190
  done
done
