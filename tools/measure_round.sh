#!/bin/bash
# Final measurements of a round, on the GPU box: bench lines (c1, c2, c3), rocprofv3 kernel stats of the
# bench command, the two HBM PMC passes of the forward, and the backward benchmark.  Everything lands in
# gpurun_out/measure/ (small files only); copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure; mkdir -p $O
for c in c1 c2; do python3 bench.py --config $c --steps 50 --warmup 10 2>/dev/null | tail -1 > $O/bench_$c.json; done
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c3.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --steps 20 --warmup 5 --cpu-budget 0 > $O/bench_c3_under_rocprof.log 2>&1
head -1 /tmp/prof_bench/*/*kernel_stats.csv > $O/kernel_stats_c3.csv
grep -E "dsph" /tmp/prof_bench/*/*kernel_stats.csv | cut -c1-300 >> $O/kernel_stats_c3.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- python3 tools/run_forward.py c3 bf16x3 fused 2 > /dev/null 2>&1
  head -1 /tmp/pmc_$ctr/*/*_counter_collection.csv > $O/pmc_$ctr.csv
  grep -E "cheb_fused_kernel|elementwise|copy" /tmp/pmc_$ctr/*/*_counter_collection.csv | cut -c1-400 >> $O/pmc_$ctr.csv
done
python3 tools/bench_backward.py c3 5 2>/dev/null | grep "^{" > $O/bench_backward_c3.json
python3 tools/bench_backward.py c2 10 2>/dev/null | grep "^{" > $O/bench_backward_c2.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bwd -- python3 tools/bench_backward.py c3 5 > /dev/null 2>&1
head -1 /tmp/prof_bwd/*/*kernel_stats.csv > $O/kernel_stats_backward_c3.csv
grep dsph /tmp/prof_bwd/*/*kernel_stats.csv | cut -c1-300 >> $O/kernel_stats_backward_c3.csv
ls -la $O
