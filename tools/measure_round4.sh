#!/bin/bash
# Round-4 measurements on the GPU box: bench lines (c1..c5 at the layers' default precision, c3 at bf16x6, the side configs
# k10 and in1), rocprofv3 kernel stats of the default bench command, SQ / GRBM counter passes of c3, and the HBM counter passes
# (one --pmc set per run, the program directly behind `--`) of tools/run_forward.py for the configs whose kernels changed this
# round (c2, c1, k10, in1) and for c3.  Small files into gpurun_out/measure4/; tools/fold_round4.py folds them into profiles/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure4; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c3.json
for c in c1 c2 k10 in1 knn8 knn20 qs qs1; do python3 bench.py --config $c --steps 100 --warmup 20 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
for c in knn20 qs qs1; do python3 bench.py --config $c --tstep off --steps 10 --warmup 3 --cpu-budget 0 --quick 2>/dev/null | tail -1 > $O/bench_${c}_gather.json; done
python3 bench.py --config knn8 --algo unfused --steps 20 --warmup 5 --cpu-budget 0 --quick 2>/dev/null | tail -1 > $O/bench_knn8_unfused.json
for c in c4 c5; do python3 bench.py --config $c --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
python3 bench.py --precision bf16x6 --steps 10 --warmup 3 --cpu-budget 0 --quick 2>/dev/null | tail -1 > $O/bench_c3_bf16x6.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --steps 20 --warmup 5 --cpu-budget 0 > $O/bench_c3_under_rocprof.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
head -1 $f > $O/kernel_stats_c3.csv; grep -E "dsph" $f | cut -c1-300 >> $O/kernel_stats_c3.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench2 -- python3 bench.py --config c2 --steps 50 --warmup 5 --cpu-budget 0 --quick > $O/bench_c2_under_rocprof.log 2>&1
f=$(find /tmp/prof_bench2 -name "*kernel_stats.csv" | head -1)
head -1 $f > $O/kernel_stats_c2.csv; grep -E "dsph" $f | cut -c1-300 >> $O/kernel_stats_c2.csv
for c in in1 knn20 qs1 c1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python3 bench.py --config $c --steps 50 --warmup 5 --cpu-budget 0 --quick > $O/bench_${c}_under_rocprof.log 2>&1
  f=$(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1)
  head -1 $f > $O/kernel_stats_$c.csv; grep -E "dsph" $f | cut -c1-300 >> $O/kernel_stats_$c.csv
done
tools/pmc3.sh r4_c3 c3 bf16x3 > $O/pmc_c3_bf16x3.txt 2>&1
PMC_ONLY="2 3 4 6" tools/pmc3.sh r4_c2 c2 bf16x3 > $O/pmc_c2_bf16x3.txt 2>&1
PMC_ONLY="2 3 4" tools/pmc3.sh r4_in1 in1 bf16x6 > $O/pmc_in1_bf16x6.txt 2>&1
PMC_ONLY="3 4" tools/pmc3.sh r4_k10 k10 bf16x6 > $O/pmc_k10_bf16x6.txt 2>&1
PMC_ONLY="3 4" tools/pmc3.sh r4_c1 c1 bf16x6 > $O/pmc_c1_bf16x6.txt 2>&1
for t in c3 c2 in1 k10 c1; do cp gpurun_out/pmc_r4_$t/summary.json $O/pmc_$t.json; rm -rf gpurun_out/pmc_r4_$t/p[0-9]*; done
ls -la $O
