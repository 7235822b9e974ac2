#!/bin/bash
# register / spill report of the structured-tile kernel instantiations (cross-compiles, no GPU needed)
cd "$(dirname "$0")/../deepsphere-cosmo-tf2_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function -fno-slp-vectorize $EXTRA -c cheb_struct.hip -o /tmp/sb/cheb_struct.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name: _ZN4dsph18cheb_struct|VGPRs:|VGPRs Spill|ScratchSize|SGPRs Spill" | grep -A4 "cheb_struct_kernel" | sed 's/.*remark: *//' | paste - - - - - | sed 's/\[-Rpass[^]]*\]//g'
