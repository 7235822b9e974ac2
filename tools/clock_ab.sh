#!/bin/bash
# GRBM_GUI_ACTIVE and duration of the strip kernel for each build_ab/sp_*.so
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
L=deepsphere-cosmo-tf2_amd/deepsphere/_lib/libdsphere_hip.so
cp $L /tmp/keep.so
export TMPDIR=/tmp
for v in build_ab/sp_*.so; do
  cp "$v" $L
  out=/tmp/clk_$(basename $v .so); rm -rf $out
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 $R/tools/run_forward.py c3 bf16x3 fused 6 > $out.log 2>&1)
  echo "== $v"
  python3 - <<PY
import csv, glob
cc = glob.glob("$out/**/*counter_collection.csv", recursive=True)
kt = glob.glob("$out/**/*kernel_trace.csv", recursive=True)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        if "strip5" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for f in cc:
    for r in csv.DictReader(open(f)):
        if "strip5" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            d = dur.get(r["Dispatch_Id"])
            c = float(r["Counter_Value"]) / 8
            print("   dispatch %s: %.0f us, %.3e cycles per XCD -> %.3f GHz" % (r["Dispatch_Id"], d or -1, c, c / (d or 1) / 1e3))
PY
done
cp /tmp/keep.so $L
