#!/bin/bash
# usage (GPU box): tools/pmc.sh <tag> <cfg> <prec> -- two SQ counter passes over tools/run_forward.py, summary per kernel
cd /tmp && export TMPDIR=/tmp
tag=$1; cfg=${2:-c3}; prec=${3:-bf16x3}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/run_forward.py $cfg $prec fused 2 > $out/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in (1, 2):
    files = glob.glob("$out/p%d/**/*counter_collection.csv" % i, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "cheb" not in k: continue
        print(k)
        for c, v in sorted(d.items()):
            print("   %-34s %.4g per dispatch" % (c, v / max(1, n[(k, c)])))
PY
