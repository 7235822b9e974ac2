#!/bin/bash
# usage (GPU box): tools/ab2.sh -- times every build_ab/v_*.so on c3 (bf16x3 and fp32), same box, two rounds
cd "$(dirname "$0")/.." || exit 1
L=deepsphere-cosmo-tf2_amd/deepsphere/_lib/libdsphere_hip.so
for round in 1 2; do
  for v in build_ab/v_*.so; do
    cp "$v" $L
    python - <<PY
import sys, os
sys.argv = ["x", "time"]
sys.path.insert(0, "tools")
import check_struct as cs
cs.timing(1024, 5, 64, 64, 4, "bf16x3", reps=8); cs.timing(1024, 5, 64, 64, 4, "fp32", reps=4)
PY
    echo "   ^ $round $v"
  done
done 2>&1 | grep -E "TIMING|\^"
