#!/bin/bash
# Round-6 measurements on the GPU box, all in one lease (one box, one clock): the bench lines (c3 -- default, strip pairs, f16 --,
# c1, c2, c4, c5 and the side configs incl. knn8h), rocprofv3 kernel stats of the default bench command and of c4, the SQ / GRBM /
# HBM counter passes of c3 and the HBM passes of c4, c5, c2, c1, k10, in1 (one --pmc set per run, the program directly behind `--`)
# over tools/run_forward.py.  Small files into gpurun_out/measure6/; tools/fold_round6.py folds them into profiles/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure6; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c3.json
python3 bench.py --steps 20 --warmup 5 --quick --strip-form pairs 2>/dev/null | tail -1 > $O/bench_c3_pairs.json
python3 bench.py --steps 20 --warmup 5 --quick --precision f16x3 2>/dev/null | tail -1 > $O/bench_c3_f16x3.json
if [ -z "$R6_C3_ONLY" ]; then
  for c in c1 c2 k10 in1 knn8 knn20; do python3 bench.py --config $c --steps 100 --warmup 20 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
  for c in c4 c5 knn8h; do python3 bench.py --config $c --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
  python3 tools/bench_backward.py > $O/bench_backward_c3.json 2>/dev/null
  python3 tools/bench_backward.py c5 > $O/bench_backward_c5.json 2>/dev/null
  python3 tools/bench_net.py 8 > $O/bench_net.json 2>/dev/null
  python3 tools/bench_net.py 8 bn > $O/bench_net_bn.json 2>/dev/null
fi
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --steps 20 --warmup 5 --cpu-budget 0 --quick > $O/bench_c3_under_rocprof.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
head -1 $f > $O/kernel_stats_c3.csv; grep -E "dsph" $f | cut -c1-300 >> $O/kernel_stats_c3.csv
if [ -z "$R6_C3_ONLY" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench4 -- python3 bench.py --config c4 --steps 10 --warmup 3 --cpu-budget 0 --quick > $O/bench_c4_under_rocprof.log 2>&1
  f=$(find /tmp/prof_bench4 -name "*kernel_stats.csv" | head -1)
  head -1 $f > $O/kernel_stats_c4.csv; grep -E "dsph" $f | cut -c1-300 >> $O/kernel_stats_c4.csv
fi
tools/pmc3.sh r6_c3 c3 bf16x3 > $O/pmc_c3_bf16x3.txt 2>&1
cp gpurun_out/pmc_r6_c3/summary.json $O/pmc_c3.json; rm -rf gpurun_out/pmc_r6_c3/p[0-9]*
if [ -z "$R6_C3_ONLY" ]; then
  PMC_ONLY="2 3 4" PMC_TIMEOUT=900 tools/pmc3.sh r6_c4 c4 bf16x3 > $O/pmc_c4_bf16x3.txt 2>&1
  PMC_ONLY="2 3 4" tools/pmc3.sh r6_c5 c5 bf16x3 > $O/pmc_c5_bf16x3.txt 2>&1
  PMC_ONLY="3 4" tools/pmc3.sh r6_c2 c2 bf16x3 > $O/pmc_c2_bf16x3.txt 2>&1
  PMC_ONLY="3 4" tools/pmc3.sh r6_c1 c1 bf16x6 > $O/pmc_c1_bf16x6.txt 2>&1
  PMC_ONLY="3 4" tools/pmc3.sh r6_k10 k10 bf16x6 > $O/pmc_k10_bf16x6.txt 2>&1
  PMC_ONLY="3 4" tools/pmc3.sh r6_in1 in1 bf16x6 > $O/pmc_in1_bf16x6.txt 2>&1
  for t in c4 c5 c2 c1 k10 in1; do cp gpurun_out/pmc_r6_$t/summary.json $O/pmc_$t.json; rm -rf gpurun_out/pmc_r6_$t/p[0-9]*; done
fi
ls -la $O
