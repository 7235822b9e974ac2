#!/usr/bin/env python3
"""Folds the per-kernel FETCH_SIZE / WRITE_SIZE means of tools/pmc3.sh (gpurun_out/measure3/pmc_<tag>.json) into
profiles/hbm_traffic.json, one key per (config, precision) as bench.py looks them up.  The round-2 entry of c3 is kept
under ":r2".  hbm_bytes_per_forward = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the kernels of one forward (the
factor 2: see the _note of the file)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "measure3")
DST = os.path.join(ROOT, "profiles", "hbm_traffic.json")
TAGS = {  # tag of the pass -> (key, note)
    "c3": ("c3:bf16x3:fused:1", "round-3 final build: strip kernel + leftover class-R tiles + face-border tiles"),
    "c3x6": ("c3:bf16x6:fused:1", "round-3 final build, bf16x6: structured tiles (class R, class T) + face-border tiles"),
    "c5": ("c5:bf16x3:fused:1", "round-3 final build, partial sky: structured tiles + border tiles (the cost gate keeps the strips off)"),
    "c4": ("c4:bf16x3:fused:1", "round-3 final build, K = 8: gather-table kernel on every tile"),
    "c2": ("c2:bf16x3:fused:1", "round-3 final build"),
    "c1": ("c1:bf16x6:fused:1", "round-3 final build"),
}
FORWARD = ("cheb_strip5_kernel", "cheb_strip_kernel", "cheb_struct_kernel", "cheb_fused_kernel")

rec = json.load(open(DST))
if "c3:bf16x3:fused:1" in rec and "c3:bf16x3:fused:1:r2" not in rec and "round-2" in rec["c3:bf16x3:fused:1"].get("kernel", ""):
    rec["c3:bf16x3:fused:1:r2"] = rec["c3:bf16x3:fused:1"]
for tag, (key, note) in TAGS.items():
    p = os.path.join(SRC, "pmc_%s.json" % tag)
    if not os.path.exists(p):
        print("missing", p, file=sys.stderr)
        continue
    d = json.load(open(p))
    fetch, write = {}, {}
    for k, v in d.items():
        if not any(f in k for f in FORWARD):
            continue
        name = k.split("(")[0].strip()
        fetch[name] = v.get("FETCH_SIZE")
        write[name] = v.get("WRITE_SIZE")
    if not fetch or any(fetch[k] is None or write[k] is None for k in fetch):
        print("incomplete passes for", tag, "- entry left as it is", file=sys.stderr)
        continue
    total = int(sum(2 * fetch[k] + write[k] for k in fetch) * 1024)
    rec[key] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_bytes_per_forward": total, "kernel": note,
                "source": "tools/pmc3.sh passes 3 and 4 (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 tools/run_forward.py), mean per dispatch"}
    print(key, total)
rec["_note"] = rec["_note"].split(" Round 3:")[0] + " Round 3: entries per config from tools/measure_round3.sh / tools/pmc3.sh via tools/traffic_from_pmc.py; every kernel of the forward is summed."
json.dump(rec, open(DST, "w"), indent=1)
