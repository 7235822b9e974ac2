#!/bin/bash
# Round-3 measurements on the GPU box: bench lines (c1..c5, at the layers' default precision, plus c3 at bf16x6 / fp32), rocprofv3
# kernel stats of the default bench command, SQ counter passes of the strip kernel, and the HBM counter passes (one --pmc set per
# run, the program directly behind `--`) of tools/run_forward.py for c3 (default precision and bf16x6), c4 and c5.
# Small files into gpurun_out/measure3/; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure3; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c3.json
for c in c1 c2; do python3 bench.py --config $c --steps 50 --warmup 10 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
for c in c4 c5; do python3 bench.py --config $c --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
python3 bench.py --precision bf16x6 --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_c3_bf16x6.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --steps 20 --warmup 5 --cpu-budget 0 > $O/bench_c3_under_rocprof.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
head -1 $f > $O/kernel_stats_c3.csv; grep -E "dsph" $f | cut -c1-300 >> $O/kernel_stats_c3.csv
tools/pmc3.sh r3_c3 c3 bf16x3 > $O/pmc_c3_bf16x3.txt 2>&1
PMC_ONLY="3 4" tools/pmc3.sh r3_c3x6 c3 bf16x6 > $O/pmc_c3_bf16x6.txt 2>&1
PMC_ONLY="3 4" tools/pmc3.sh r3_c5 c5 bf16x3 > $O/pmc_c5_bf16x3.txt 2>&1
PMC_ONLY="3 4" tools/pmc3.sh r3_c4 c4 bf16x3 > $O/pmc_c4_bf16x3.txt 2>&1
PMC_ONLY="3 4" tools/pmc3.sh r3_c2 c2 bf16x3 > $O/pmc_c2_bf16x3.txt 2>&1
PMC_ONLY="3 4" tools/pmc3.sh r3_c1 c1 bf16x6 > $O/pmc_c1_bf16x6.txt 2>&1
for t in c3 c3x6 c5 c4 c2 c1; do cp gpurun_out/pmc_r3_$t/summary.json $O/pmc_$t.json; rm -rf gpurun_out/pmc_r3_$t/p[0-9]*; done
ls -la $O
