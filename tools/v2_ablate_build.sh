#!/bin/bash
# Ablation builds of libevk.so (timing only; results are wrong below the last stage), loaded with EVK_LIB_PATH:
#   -DV2_ABLATE_A=1|2|3   one-pass partition stops after the histogram / the scan / the placement (evk_voxel2.hip)
#   -DV2_ABLATE_B=0|1|2   voxel tile kernel stops after the table entries / the record loads / the decode
#   -DIWE_ABLATE=0|1      tiled IWE kernel: record loads only / + per-event arithmetic without LDS atomics (evk_tiled.hip)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ablate
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -ldl"
for a in 1 2 3; do
  /opt/rocm/bin/hipcc $FLAGS -DV2_ABLATE_A=$a event_utils_amd/csrc/*.hip -o tools/ablate/libevk_a$a.so &
done
for b in 0 1 2; do
  /opt/rocm/bin/hipcc $FLAGS -DV2_ABLATE_B=$b event_utils_amd/csrc/*.hip -o tools/ablate/libevk_b$b.so &
done
for i in 0 1; do
  /opt/rocm/bin/hipcc $FLAGS -DIWE_ABLATE=$i event_utils_amd/csrc/*.hip -o tools/ablate/libevk_iwe$i.so &
done
wait
ls tools/ablate
