#!/bin/bash
# usage (GPU box): tools/stamps_strip.sh  -- s_memtime shares of sixteen steps of one workgroup, for every build_ab/sp_*s.so
cd "$(dirname "$0")/.." || exit 1
L=deepsphere-cosmo-tf2_amd/deepsphere/_lib/libdsphere_hip.so
cp $L /tmp/keep.so
for v in build_ab/sp_*s.so; do
  cp "$v" $L
  echo "== $v"
  DSPH_STAMPS_DUMP=1 python3 tools/run_forward.py c3 bf16x3 fused 2 2>&1 | grep SPSTAMP | tail -32
done
cp /tmp/keep.so $L
