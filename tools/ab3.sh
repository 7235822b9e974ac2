#!/bin/bash
# usage (GPU box): tools/ab3.sh -- times every build_ab/v_*.so on c3 in all three contraction arithmetics, same box
cd "$(dirname "$0")/.." || exit 1
L=deepsphere-cosmo-tf2_amd/deepsphere/_lib/libdsphere_hip.so
for round in 1 2; do
  for v in build_ab/v_*.so; do
    cp "$v" $L
    python - <<PY
import sys
sys.argv = ["x", "time"]
sys.path.insert(0, "tools")
import check_struct as cs
for p in ("bf16x3", "bf16x6", "fp32"):
    cs.timing(1024, 5, 64, 64, 4, p, reps=6)
PY
    echo "   ^ $round $v"
  done
done 2>&1 | grep -E "TIMING|\^"
