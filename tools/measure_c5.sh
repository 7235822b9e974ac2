#!/bin/bash
# configs[4] after its mask-edge tiles left the breadth-first kernel (round 6's last build): HBM counter passes, then the bench lines
# of c5 and c5s so that they carry the refreshed entry.  tools/fold_round6.py c5 folds gpurun_out/measure6/pmc_c5.json.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure6; mkdir -p $O
PMC_TIMEOUT=600 PMC_ONLY="2 3 4" tools/pmc3.sh r6_c5 c5 bf16x3 > $O/pmc_c5_bf16x3.txt 2>&1
cp gpurun_out/pmc_r6_c5/summary.json $O/pmc_c5.json; rm -rf gpurun_out/pmc_r6_c5/p[0-9]*
python3 bench.py --config c5 --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_c5.json
grep -A 12 "qstrip5\|cheb_struct_kernel\|cheb_fused_kernel" $O/pmc_c5_bf16x3.txt | grep -E "dsph|FETCH|WRITE|INSTS_VALU" | head -20
