#!/bin/bash
# usage (GPU box): tools/ab_strip.sh [cfg]  -- average time of the strip kernel for every build_ab/sp_*.so (rocprofv3 kernel stats)
cd "$(dirname "$0")/.." || exit 1
R=$PWD
cfg=${1:-c3}
L=deepsphere-cosmo-tf2_amd/deepsphere/_lib/libdsphere_hip.so
cp $L /tmp/keep.so
export TMPDIR=/tmp
for v in build_ab/sp_*.so; do
  cp "$v" $L
  out=/tmp/prof_$(basename $v .so)
  rm -rf $out
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/tools/run_forward.py $cfg bf16x3 fused 6 > $out.log 2>&1)
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  grep -E "strip|struct_kernel|fused_kernel" $f | awk -F, '{printf "   %-60s calls %s avg_ns %s min_ns %s\n", substr($1,1,60), $2, $4, $6}'
done
cp /tmp/keep.so $L
