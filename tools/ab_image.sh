#!/bin/bash
# A/B of builds of libevk.so on the SAME box for the event-image path: tools/ab_image.sh name1 name2 ... ("default" = product)
for i in 1 2; do
  for name in "$@"; do
    lib=tools/exp/libevk_$name.so; [ "$name" = default ] && lib=event_utils_amd/csrc/libevk.so
    echo "== $name"; EVK_LIB_PATH=$PWD/$lib python tools/image2_time.py ${ROTATE:+--rotate} 2>&1 | grep "n= 10000000" | grep "int32\|bilinear unit" | cut -c1-150
  done
done
