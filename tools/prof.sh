#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof.sh <tag> [bench args]  -> gpurun_out/prof_<tag>/ kernel stats
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-budget 0 "$@" > $out/bench.log 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
head -8 $out/kernel_stats.csv
tail -2 $out/bench.log | cut -c1-600
