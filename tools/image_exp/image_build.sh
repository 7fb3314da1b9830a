#!/bin/bash
# Round 6's experiment: a persistent raw range image per scene.  Builds the library with the
# experiment's workspace fields and kernels (-DR3D_EXP_IMAGE: k_image_clear, k_image_build, k_image_bands of
# tools/image_exp/r3d_image.hip behind r3d_batch_launch_one ids 6 / 7 / 8) as libreal3daug_hip_image.so; to time them:
#   bash tools/image_exp/image_build.sh && R3D_LIB=pcl-augmentation_amd/libreal3daug_hip_image.so python tools/image_exp/image_build.py C2
# Results: profiles/r06_image_build.md.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/pcl-augmentation_amd/csrc"
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -Wno-pass-failed -DR3D_EXP_IMAGE"
pids=()
for f in r3d_level1 r3d_batch r3d_places r3d_richmap; do hipcc $FLAGS -c $f.hip -o $T/$f.o & pids+=($!); done
hipcc $FLAGS -mllvm -disable-machine-licm -c r3d_insert.hip -o $T/r3d_insert.o & pids+=($!)
hipcc $FLAGS -mllvm -disable-machine-licm -I. -c "$R/tools/image_exp/r3d_image.hip" -o $T/r3d_image.o & pids+=($!)
hipcc $FLAGS -c r3d_hostpack.cpp -o $T/r3d_hostpack.o & pids+=($!)
for p in "${pids[@]}"; do wait $p || { echo "image_build: a compile job failed" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC $T/*.o -lpthread -o ../libreal3daug_hip_image.so
ls -la ../libreal3daug_hip_image.so
