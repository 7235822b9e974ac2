#!/bin/bash
# usage: tools/gpu.sh <timeout-s> '<remote command>'   (always from the repo root; rebuilds the library first)
cd "$(dirname "$0")/.." || exit 1
make -C deepsphere-cosmo-tf2_amd/csrc -j8 ${MAKEFLAGS_EXTRA} 2>&1 | grep -E "error|warning: v|spill" | head
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
