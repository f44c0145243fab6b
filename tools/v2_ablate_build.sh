#!/bin/bash
# Ablation builds of libevk.so for the one-pass voxel path (timing only; see V2_ABLATE_* in evk_voxel2.hip).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ablate
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function"
for a in 1 2 3; do
  /opt/rocm/bin/hipcc $FLAGS -DV2_ABLATE_A=$a event_utils_amd/csrc/*.hip -o tools/ablate/libevk_a$a.so &
done
for b in 0 1 2; do
  /opt/rocm/bin/hipcc $FLAGS -DV2_ABLATE_B=$b event_utils_amd/csrc/*.hip -o tools/ablate/libevk_b$b.so &
done
wait
ls tools/ablate
