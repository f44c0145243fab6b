#!/bin/bash
# Experiments build of libevk.so (-DEVK_EXPERIMENTS: the partition geometries of evk_voxel2.hip that the product does not
# ship, selected with EVK_V2_PART), loaded with EVK_LIB_PATH=tools/exp/libevk_exp.so.  Extra -D flags may be passed.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -ldl \
  -DEVK_EXPERIMENTS "$@" event_utils_amd/csrc/*.hip -o tools/exp/libevk_exp.so
ls -la tools/exp/libevk_exp.so
