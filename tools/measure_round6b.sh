#!/bin/bash
# Round 6, second lease, FINAL build: the bench lines again so that they carry this round's traffic entries (tools/fold_round6.py
# refreshed profiles/hbm_traffic.json from the first lease's counters) and the K = 8 kernel's eight-wave variant; kernel stats of
# the default bench command and of c4.  Small files into gpurun_out/measure6/ (the counter summaries of the first lease stay).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure6; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c3.json
python3 bench.py --steps 20 --warmup 5 --quick --strip-form pairs 2>/dev/null | tail -1 > $O/bench_c3_pairs.json
python3 bench.py --steps 20 --warmup 5 --quick --precision f16x3 2>/dev/null | tail -1 > $O/bench_c3_f16x3.json
for c in c1 c2 k10 in1; do python3 bench.py --config $c --steps 100 --warmup 20 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
for c in c4 c5 knn8h; do python3 bench.py --config $c --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --steps 20 --warmup 5 --cpu-budget 0 --quick > $O/bench_c3_under_rocprof.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
head -1 $f > $O/kernel_stats_c3.csv; grep -E "dsph" $f | cut -c1-300 >> $O/kernel_stats_c3.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench4 -- python3 bench.py --config c4 --steps 10 --warmup 3 --cpu-budget 0 --quick > $O/bench_c4_under_rocprof.log 2>&1
f=$(find /tmp/prof_bench4 -name "*kernel_stats.csv" | head -1)
head -1 $f > $O/kernel_stats_c4.csv; grep -E "dsph" $f | cut -c1-300 >> $O/kernel_stats_c4.csv
ls -la $O | head -40
