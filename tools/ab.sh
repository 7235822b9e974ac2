#!/bin/bash
# A/B on one GPU box: times bench.py with every library variant in build_ab/*.so (same box, back to back, twice).
cd "$(dirname "$0")/.." || exit 1
L=deepsphere-cosmo-tf2_amd/deepsphere/_lib/libdsphere_hip.so
cp $L /tmp/orig.so
for round in 1 2; do
  for v in build_ab/*.so; do
    cp "$v" $L
    ms=$(python bench.py --steps ${AB_STEPS:-20} --warmup 5 --cpu-budget 0 --precision ${AB_PREC:-bf16x3} 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "$round $v $ms"
  done
done
cp /tmp/orig.so $L
