#!/bin/bash
# an experimental build of the library: tools/build_variant.sh <name> <extra hipcc flags...>  ->  pcl-augmentation_amd/libreal3daug_hip_<name>.so
# (the other objects are taken from the normal build: only r3d_insert.hip and r3d_batch.hip are recompiled with the flags)
set -e
name=$1; shift
cd "$(dirname "$0")/../pcl-augmentation_amd/csrc"
T=$(mktemp -d)
for f in r3d_batch r3d_insert; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -Wno-pass-failed "$@" $([ $f = r3d_insert ] && echo "-mllvm -disable-machine-licm") -c $f.hip -o $T/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $T/r3d_batch.o $T/r3d_insert.o r3d_level1.o r3d_places.o r3d_richmap.o r3d_hostpack.o -lpthread -o ../libreal3daug_hip_$name.so
ls -la ../libreal3daug_hip_$name.so
