#!/bin/bash
# diagnostic build of the library with index checks in the insert / compaction kernels (-DR3D_CHECK; tools/check_soak.py)
set -e
cd "$(dirname "$0")/../pcl-augmentation_amd/csrc"
T=$(mktemp -d)
for f in r3d_level1 r3d_batch r3d_insert r3d_places r3d_richmap; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -Wno-pass-failed -DR3D_CHECK $([ $f = r3d_insert ] && echo "-mllvm -disable-machine-licm") -c $f.hip -o $T/$f.o &
done
wait
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c r3d_hostpack.cpp -o $T/r3d_hostpack.o
hipcc --offload-arch=gfx950 -shared -fPIC $T/*.o -lpthread -o ../libreal3daug_hip_check.so
