#!/bin/bash
# an experimental build of the library: tools/build_variant.sh <name> <extra hipcc flags...>  ->  pcl-augmentation_amd/libreal3daug_hip_<name>.so
exec bash "$(dirname "$0")/build_flavour.sh" "$@"
