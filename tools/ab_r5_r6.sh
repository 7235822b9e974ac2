#!/bin/bash
# One box, one clock: the headline forward on round 5's tree (built by hand into build_ab/r5: `git archive <round-5 commit> |
# tar -x -C build_ab/r5; make -C build_ab/r5/deepsphere-cosmo-tf2_amd/csrc`) and on this tree, alternating; and this tree with
# the class-T tiles kept off the strips (DSPH_QT_ONLY_R: round 5's tile set on this round's kernel; honoured by a
# `make ABLATE=1` build of the library only -- the shipped one reads no environment variable, that leg then equals the second).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/ab56; mkdir -p $O
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"; }
for i in 1 2 3; do
  (cd build_ab/r5 && python3 bench.py --steps 20 --warmup 5 --quick --cpu-budget 0 2>/dev/null | tail -1 | line r5) >> $O/ab.txt
  python3 bench.py --steps 20 --warmup 5 --quick --cpu-budget 0 2>/dev/null | tail -1 | line r6 >> $O/ab.txt
  DSPH_QT_ONLY_R=1 python3 bench.py --steps 20 --warmup 5 --quick --cpu-budget 0 2>/dev/null | tail -1 | line r6_only_r >> $O/ab.txt
done
cat $O/ab.txt
for i in 1 2; do
  python3 bench.py --config c1 --steps 200 --warmup 50 --quick --cpu-budget 0 2>/dev/null | tail -1 | line c1 >> $O/ab.txt
  python3 bench.py --config c1 --steps 200 --warmup 50 --quick --cpu-budget 0 --struct off 2>/dev/null | tail -1 | line c1_bfs_only >> $O/ab.txt
done
tail -4 $O/ab.txt
