#!/bin/bash
# Round-5 measurements on the GPU box, c3 (the headline): the bench line with both strip forms on the same box, rocprofv3 kernel
# stats of the default bench command, the SQ / GRBM / HBM counter passes (one --pmc set per run, the program directly behind
# `--`) of tools/run_forward.py.  Small files into gpurun_out/measure5/; tools/fold_round5.py folds them into profiles/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure5; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c3.json
python3 bench.py --steps 20 --warmup 5 --quick --strip-form pairs 2>/dev/null | tail -1 > $O/bench_c3_pairs.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --steps 20 --warmup 5 --cpu-budget 0 > $O/bench_c3_under_rocprof.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
head -1 $f > $O/kernel_stats_c3.csv; grep -E "dsph" $f | cut -c1-300 >> $O/kernel_stats_c3.csv
tools/pmc3.sh r5_c3 c3 bf16x3 > $O/pmc_c3_bf16x3.txt 2>&1
cp gpurun_out/pmc_r5_c3/summary.json $O/pmc_c3.json; rm -rf gpurun_out/pmc_r5_c3/p[0-9]*
ls -la $O
