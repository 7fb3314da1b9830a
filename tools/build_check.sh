#!/bin/bash
# diagnostic build of the library with index checks in the insert / compaction kernels (-DR3D_CHECK; tools/check_soak.py)
exec bash "$(dirname "$0")/build_flavour.sh" check -DR3D_CHECK
