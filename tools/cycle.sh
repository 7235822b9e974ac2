#!/bin/bash
# usage (GPU box): tools/cycle.sh -- parity + timings with the in-tree library, then the stamp deltas of build_ab/stamps.so
cd "$(dirname "$0")/.." || exit 1
python tools/check_struct.py ${1:-all} 2>&1 | grep -v amdgpu.ids | tail -14
if [ -f build_ab/stamps.so ]; then
  L=deepsphere-cosmo-tf2_amd/deepsphere/_lib/libdsphere_hip.so
  cp build_ab/stamps.so $L
  DSPH_STAMPS_DUMP=1 python tools/run_forward.py c3 ${2:-bf16x3} fused 2 2>&1 | grep -E "STSTAMP wave (0|4|7) item (5|6)"
fi
