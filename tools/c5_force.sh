#!/bin/bash
# usage (GPU box): tools/c5_force.sh -- BASELINE configs[4] (partial sky) with the strip kernel forced on and as the cost gate leaves it
cd "$GRAFT_REPO_ROOT" || exit 1
for f in 1 0; do
  if [ $f = 1 ]; then export DSPH_STRIP_FORCE=1; else unset DSPH_STRIP_FORCE; fi
  echo "c5 force=$f $(python3 bench.py --config c5 --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['dtype'][-90:])")"
done
unset DSPH_STRIP_FORCE
echo "c3 $(python3 bench.py --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['dtype'][-90:])")"
