#!/bin/bash
# K = 10 in one pass (round 6's last build): the bench line, the HBM counter passes and kernel stats of the side config k10.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure6; mkdir -p $O
PMC_ONLY="3 4" tools/pmc3.sh r6_k10b k10 bf16x3 > $O/pmc_k10_bf16x3.txt 2>&1
cp gpurun_out/pmc_r6_k10b/summary.json $O/pmc_k10b.json; rm -rf gpurun_out/pmc_r6_k10b/p[0-9]*
python3 bench.py --config k10 --steps 100 --warmup 20 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_k10.json
python3 tools/k10_routes.py > $O/k10_routes.txt 2>&1
tail -3 $O/pmc_k10_bf16x3.txt; cut -c1-300 $O/bench_k10.json; cat $O/k10_routes.txt
