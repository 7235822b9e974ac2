#!/bin/bash
# Round 5, last part (after tools/fold_round5.py): the bench lines again, so that every line's roofline.traffic is this round's
# entry of profiles/hbm_traffic.json, and forward + backward at c3 / c2.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure5; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c3.json
python3 bench.py --steps 20 --warmup 5 --quick --strip-form pairs 2>/dev/null | tail -1 > $O/bench_c3_pairs.json
python3 bench.py --steps 20 --warmup 5 --quick --precision f16x3 2>/dev/null | tail -1 > $O/bench_c3_f16x3.json
for c in c1 c2; do python3 bench.py --config $c --steps 100 --warmup 20 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
for c in c4 c5; do python3 bench.py --config $c --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
python3 tools/bench_backward.py c3 5 auto 2>/dev/null | tail -1 > $O/bench_backward_c3.json
python3 tools/bench_backward.py c2 10 auto 2>/dev/null | tail -1 > $O/bench_backward_c2.json
ls -la $O
