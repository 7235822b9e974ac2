#!/bin/bash
# usage (GPU box): tools/pmc3.sh <tag> [cfg] [prec] -- SQ + memory counter passes over tools/run_forward.py (one --pmc set per run), summary per kernel
cd /tmp && export TMPDIR=/tmp
tag=$1; cfg=${2:-c3}; prec=${3:-bf16x3}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
      "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
      "FETCH_SIZE"
      "WRITE_SIZE"
      "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
      "GRBM_GUI_ACTIVE"
      "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC")
i=0
for P in "${SETS[@]}"; do
  i=$((i+1))
  # PMC_ONLY="3 4": just those passes (FETCH_SIZE and WRITE_SIZE)
  if [ -n "$PMC_ONLY" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then continue; fi
  echo "pass $i: $P"
  timeout ${PMC_TIMEOUT:-400} rocprofv3 --pmc $P --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/run_forward.py $cfg $prec ${PMC_ALGO:-fused} 2 > $out/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
res = {}
for i in range(1, 8):
    files = glob.glob("$out/p%d/**/*counter_collection.csv" % i, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "dsph" not in k and "elementwise" not in k: continue
        for c, v in sorted(d.items()):
            res.setdefault(k, {})[c] = v / max(1, n[(k, c)])
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-34s %.5g per dispatch" % (c, v))
json.dump(res, open("$out/summary.json", "w"), indent=1)
PY
