#!/bin/bash
# usage (GPU box): tools/pmc_qwgrad.sh -- SQ / LDS / HBM counter passes (one --pmc set per run, the program directly behind `--`) over
# the weight-gradient probe's full-chip case (S = 1024, eight maps: tools/ubench/qwgrad_probe 3); per-dispatch means of
# cheb_qwgrad5_kernel into gpurun_out/pmc_qwgrad/summary.json
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_qwgrad
mkdir -p $out
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
      "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
      "FETCH_SIZE"
      "WRITE_SIZE"
      "GRBM_GUI_ACTIVE")
i=0
for P in "${SETS[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $P --output-format csv -d $out/p$i -- $GRAFT_REPO_ROOT/tools/ubench/qwgrad_probe 3 > $out/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
res = {}
for i in range(1, 6):
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob("$out/p%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if "cheb_qwgrad5" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c, v in acc.items(): res[c] = v / max(1, n[c])
res["_note"] = "mean per dispatch of cheb_qwgrad5_kernel<true> over tools/ubench/qwgrad_probe 3 (S = 1024, N = 8, 256 workgroups, ~578 steps each); FETCH_SIZE / WRITE_SIZE in KiB (HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE KiB on gfx950, MI355X_MICROARCH.md)"
json.dump(res, open("$out/summary.json", "w"), indent=1)
for c, v in sorted(res.items()):
    print("%-30s %s" % (c, v))
PY
rm -rf $out/p[0-9]
