#!/bin/bash
# usage (GPU box): tools/stamps.sh [cfg] -- runs build_ab/stamps.so (make STAMPS=1) on one config and prints the stamp deltas
cd "$(dirname "$0")/.." || exit 1
L=deepsphere-cosmo-tf2_amd/deepsphere/_lib/libdsphere_hip.so
cp build_ab/stamps.so $L
DSPH_STAMPS_DUMP=1 python tools/run_forward.py ${1:-c3} bf16x3 fused 2 2>&1 | grep -E "STSTAMP|done" | tail -66
