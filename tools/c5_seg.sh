#!/bin/bash
# usage (GPU box): tools/c5_seg.sh -- BASELINE configs[4] with the strip kernel forced on, for a range of segment heights
cd "$GRAFT_REPO_ROOT" || exit 1
export DSPH_STRIP_FORCE=1
for h in 64 128 256 512 4096; do
  export DSPH_STRIP_SEG=$h
  echo "c5 seg=$h $(DSPH_STRIP_DEBUG=1 python3 bench.py --config c5 --steps 10 --warmup 3 --cpu-budget 0 2>/tmp/err.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])") $(grep -c build_strips /tmp/err.log)"
done
