#!/bin/bash
# Round-2 measurements on the GPU box: bench lines (c1..c5), rocprofv3 kernel stats of the default bench command, SQ counter
# passes and the two HBM counter passes (separate --pmc runs) of tools/run_forward.py.  Small files into gpurun_out/measure2/;
# copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure2; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c3.json
for c in c1 c2; do python3 bench.py --config $c --steps 50 --warmup 10 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
for c in c4 c5; do python3 bench.py --config $c --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_$c.json; done
python3 bench.py --precision fp32 --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_c3_fp32.json
python3 bench.py --precision bf16x6 --steps 10 --warmup 3 --cpu-budget 0 2>/dev/null | tail -1 > $O/bench_c3_bf16x6.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --steps 20 --warmup 5 --cpu-budget 0 > $O/bench_c3_under_rocprof.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
head -1 $f > $O/kernel_stats_c3.csv; grep -E "dsph" $f | cut -c1-300 >> $O/kernel_stats_c3.csv
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc_$i -- python3 tools/run_forward.py c3 bf16x3 fused 2 > /dev/null 2>&1
done
python3 - <<PY > $O/pmc_summary.json
import csv, glob, collections, json
out = {}
for i in (1, 2, 3, 4):
    for f in glob.glob("/tmp/pmc_%d/**/*counter_collection.csv" % i, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            if "dsph" not in k and "elementwise" not in k and "copy" not in k.lower():
                continue
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            out.setdefault(k, {})[c] = {"per_dispatch_mean": sum(v) / len(v), "dispatches": len(v), "values": v[:4]}
print(json.dumps(out, indent=1))
PY
ls -la $O
