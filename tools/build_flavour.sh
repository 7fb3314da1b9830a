#!/bin/bash
# A diagnostic / experimental build of the whole library:
#   tools/build_flavour.sh <name> [extra hipcc flags...]  ->  pcl-augmentation_amd/libreal3daug_hip_<name>.so
# Every source is compiled afresh into a temporary directory (nothing is taken from the normal build's objects), every
# compile job is waited for by its own PID (a failed job fails the script, it does not surface as a link error), and
# r3d_level1 gets the same R3D_SRC_HASH csrc/Makefile passes, so that r3d_build_info() of the flavour names the sources it
# was built from (__graft_entry__.build() compares it with the tree when the flavour is loaded through R3D_LIB).
set -e
name=$1; shift
cd "$(dirname "$0")/../pcl-augmentation_amd/csrc"
SRCS="r3d_level1.hip r3d_batch.hip r3d_insert.hip r3d_places.hip r3d_richmap.hip r3d_hostpack.cpp"
HDRS="r3d_device.hpp r3d_host.hpp r3d_batch.hpp r3d_insert_core.hpp ../../include/real3daug_hip.h"
HASH=$(cat $SRCS $HDRS | sha256sum | cut -c1-16)
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -Wno-pass-failed"
pids=()
for f in r3d_level1 r3d_batch r3d_insert r3d_places r3d_richmap; do
  extra=()
  [ $f = r3d_insert ] && extra=(-mllvm -disable-machine-licm)
  [ $f = r3d_level1 ] && extra=("-DR3D_SRC_HASH=\"$HASH\"")
  hipcc $FLAGS "$@" "${extra[@]}" -c $f.hip -o $T/$f.o &
  pids+=($!)
done
hipcc $FLAGS -c r3d_hostpack.cpp -o $T/r3d_hostpack.o &
pids+=($!)
for p in "${pids[@]}"; do
  wait $p || { echo "build_flavour: a compile job failed" >&2; exit 1; }
done
hipcc --offload-arch=gfx950 -shared -fPIC $T/*.o -lpthread -o ../libreal3daug_hip_$name.so
ls -la ../libreal3daug_hip_$name.so
