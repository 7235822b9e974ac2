#!/bin/bash
# Round-5 measurements of the backward pass on the GPU box, one lease: forward + backward at c3 and c2 (tools/bench_backward.py),
# rocprofv3 kernel stats of the c3 command, the weight-gradient probe (checks, timing, ablation builds when present) and the error
# of dW by order at c3's size for the three weight-gradient routes (tests/diag_dw_by_order.py).  Small files into gpurun_out/measure5d/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure5d; mkdir -p $O
python3 tools/bench_backward.py c3 5 auto 2>/dev/null | tail -1 > $O/bench_backward_c3.json
python3 tools/bench_backward.py c2 10 auto 2>/dev/null | tail -1 > $O/bench_backward_c2.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bwd -- python3 tools/bench_backward.py c3 5 auto > $O/bench_backward_c3_under_rocprof.log 2>&1
f=$(find /tmp/prof_bwd -name "*kernel_stats.csv" | head -1)
head -1 $f > $O/kernel_stats_backward_c3.csv; grep -E "dsph" $f | cut -c1-300 >> $O/kernel_stats_backward_c3.csv
{ timeout 200 tools/ubench/qwgrad_probe
  for v in abl6 abl8192; do [ -x tools/ubench/qwgrad_probe_$v ] && { echo "build -DDSPH_QS_ABL=${v#abl} (timing only: results wrong by construction)"; timeout 60 tools/ubench/qwgrad_probe_$v 3; }; done
} > $O/qwgrad_probe.txt 2>&1
timeout 500 python3 tests/diag_dw_by_order.py 1024 4 2>/dev/null > $O/dw_error_by_order.txt
timeout 300 python3 tests/diag_dw_by_order.py 512 4 2>/dev/null >> $O/dw_error_by_order.txt
ls -la $O
