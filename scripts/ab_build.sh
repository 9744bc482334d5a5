#!/bin/bash
# Build an experimental variant of the library next to the normal one: scripts/ab_build.sh -DFI_SOMETHING ...
# -> feature_intertwiner_amd/libfi_hip_exp.so (git-ignored; travels with gpurun).  On the GPU box:
#   cp feature_intertwiner_amd/libfi_hip_exp.so feature_intertwiner_amd/libfi_hip.so   to run the variant.
set -e
cd "$(dirname "$0")/../feature_intertwiner_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function"
mkdir -p /tmp/fi_exp
objs=""
for s in *.hip; do
  o=/tmp/fi_exp/${s%.hip}.o
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $s -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfi_hip_exp.so $objs
ls -la ../libfi_hip_exp.so
