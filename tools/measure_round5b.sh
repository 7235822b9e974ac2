#!/bin/bash
# Round 5, second part: the HBM counter passes of c4 (its set-up alone takes minutes under the profiler) and c2.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure5; mkdir -p $O
PMC_TIMEOUT=1500 PMC_ONLY="3 4" tools/pmc3.sh r5_c4 c4 bf16x3 > $O/pmc_c4_bf16x3.txt 2>&1
PMC_ONLY="3 4" tools/pmc3.sh r5_c2 c2 bf16x3 > $O/pmc_c2_bf16x3.txt 2>&1
for t in c4 c2; do cp gpurun_out/pmc_r5_$t/summary.json $O/pmc_$t.json; tail -3 gpurun_out/pmc_r5_$t/p3.log > $O/pmc_${t}_p3_tail.log; rm -rf gpurun_out/pmc_r5_$t/p[0-9]*; done
ls -la $O
