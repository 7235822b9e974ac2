#!/bin/bash
# usage (GPU box): tools/ab_tables.sh -- same library, same box: class-T tiles on the structured kernel (default) vs on the
# BFS-tile kernel (DSPH_NO_TABLES=1) vs everything on the BFS-tile kernel (DSPH_NO_STRUCT=1), configs c2 c3 c5
cd "$(dirname "$0")/.." || exit 1
for cfg in c3 c5 c2; do
  for mode in tables notables nostruct; do
    case $mode in
      tables) env_="";;
      notables) env_="DSPH_NO_TABLES=1";;
      nostruct) env_="DSPH_NO_STRUCT=1";;
    esac
    env $env_ python bench.py --config $cfg --cpu-budget 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$cfg $mode', d['ms_per_step'], 'ms  frac', d['roofline']['frac'], d['roofline']['kernel'])"
  done
done
