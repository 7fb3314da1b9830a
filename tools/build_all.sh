#!/bin/bash
# the library and its diagnostic (stamps) build, from anywhere
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
make -C "$R/pcl-augmentation_amd/csrc" -j4 2>&1 | grep -E "error|warning" || true
bash "$R/tools/build_stamps.sh"
ls -la "$R"/pcl-augmentation_amd/*.so
