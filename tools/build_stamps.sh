#!/bin/bash
# diagnostic build of the library with per-phase stamps in the insert kernel (tools/stamps_insert.py)
exec bash "$(dirname "$0")/build_flavour.sh" stamps -DR3D_STAMPS
