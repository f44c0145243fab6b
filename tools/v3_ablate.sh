#!/bin/bash
# Ablation builds of libevk.so for the 4-byte-record voxel path (timing only): V3_ABLATE_P / V3_ABLATE_T of evk_voxel3.hip
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ablate; rm -f tools/ablate/*.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function"
for f in evk_cmax evk_comm evk_imgops evk_scatter evk_tiled evk_voxel2; do
  /opt/rocm/bin/hipcc $FLAGS -c event_utils_amd/csrc/$f.hip -o tools/ablate/$f.o &
done
wait
for v in P1 P2 P3 P4 T1 T2 T3 T4; do
  ( /opt/rocm/bin/hipcc $FLAGS -DV3_ABLATE_${v:0:1}=${v:1:1} -c event_utils_amd/csrc/evk_voxel3.hip -o tools/ablate/v3_$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/ablate/evk_*.o tools/ablate/v3_$v.o -ldl -o tools/ablate/libevk_$v.so ) &
done
wait
rm -f tools/ablate/*.o
ls tools/ablate
