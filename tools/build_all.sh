#!/bin/bash
# builds the in-tree library and the stamps variant (build_ab/stamps.so)
cd "$(dirname "$0")/.." || exit 1
make -C deepsphere-cosmo-tf2_amd/csrc -j8 $MK 2>&1 | grep -E "error|warning: v" 
make -C deepsphere-cosmo-tf2_amd/csrc -j8 STAMPS=1 OBJDIR=/tmp/sb/stamps OUTDIR=/tmp/sb/stamps_lib $MK 2>&1 | grep -E "error|warning: v"
mkdir -p build_ab && cp /tmp/sb/stamps_lib/libdsphere_hip.so build_ab/stamps.so
