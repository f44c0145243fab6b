#!/bin/bash
# A/B builds of libevk.so: tools/exp/libevk_<name>.so for every "name:flags" argument, e.g.
#   tools/ab_build.sh "ieee:-DEVK_AB_IEEE_DIV" "magic:-DEVK_AB_MAGIC_KEY"
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/exp
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -ldl \
    $flags event_utils_amd/csrc/*.hip -o tools/exp/libevk_$name.so &
done
wait
ls tools/exp
