#!/bin/bash
# LDS counters of the strip kernel for each build_ab/sp_*.so
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
L=deepsphere-cosmo-tf2_amd/deepsphere/_lib/libdsphere_hip.so
cp $L /tmp/keep.so
export TMPDIR=/tmp
for v in build_ab/sp_*.so; do
  cp "$v" $L
  out=/tmp/clk_$(basename $v .so); rm -rf $out
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 $R/tools/run_forward.py c3 bf16x3 fused 6 > $out.log 2>&1)
  echo "== $v"
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "strip5" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("   %-24s %.4e (mean of %d)" % (k, sum(acc[k]) / len(acc[k]), len(acc[k])))
PY
done
cp /tmp/keep.so $L
