#!/bin/bash
# needs a diagnostic library: make -C deepsphere-cosmo-tf2_amd/csrc clean && make -C deepsphere-cosmo-tf2_amd/csrc -j8 ABLATE=1
# timing-only ablations of the fused kernel (outputs are wrong when a bit is set)
for d in 0 1 2 3 8 9 10 11; do  # bits: 1 no recurrence, 2 no contraction, 8 no y store
  echo -n "dbg=$d  "
  DSPH_FUSED_DEBUG=$d python bench.py --steps 5 --warmup 1 --cpu-budget 0 --precision ${1:-bf16x3} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"
done
