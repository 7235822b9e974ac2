#!/usr/bin/env python3
"""Folds gpurun_out/measure6/ (tools/measure_round6.sh) into profiles/: the bench lines and kernel stats as r6_* files, the
counter summaries as profiles/r6_fused_pmc.json (one entry per config: kernel -> counter -> mean per dispatch), and the
FETCH_SIZE / WRITE_SIZE means into profiles/hbm_traffic.json under the keys bench.py looks up -- with the kernel mix of the
forward they were measured on (`kernels`: the string bench.py prints as roofline.kernel) and the commit of the build.
hbm_bytes_per_forward = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the kernels of one forward (the factor 2: see the
_note of that file); issue counts (SQ_INSTS_VALU, SQ_INSTS_MFMA summed over the kernels of a forward) go in next to it."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "measure6")
PRO = os.path.join(ROOT, "profiles")
KEYS = {"c3": "c3:bf16x3:fused:1", "c5": "c5:bf16x3:fused:1", "c4": "c4:bf16x3:fused:1", "c1": "c1:bf16x6:fused:1", "c2": "c2:bf16x3:fused:1",
        "k10": "k10:bf16x6:fused:1", "in1": "in1:bf16x6:fused:1",
        "k10b": "k10:bf16x3:fused:1"}  # (k10b: K = 10 in one pass of the breadth-first tile kernel, the default route since round 6's last build)
ONLY = sys.argv[1:]  # tags to fold (default: all): `fold_round6.py k10b` adds one entry and leaves the others' build stamps alone
FORWARD = ("cheb_qstrip5_kernel", "cheb_qstrip8_kernel", "cheb_strip5_kernel", "cheb_strip_kernel", "cheb_istrip_kernel", "cheb_istrip1_kernel", "cheb_struct_kernel",
           "cheb_fused_kernel", "fused_pad_kernel")
# what the issue model of bench.py needs to know about the dominant kernel of a config (static: read off the kernel source)
MODEL = {"c3": {"dpp_share_of_valu": 0.13, "mfma_pipe_cycles_each": 16, "waves_per_simd": 2},
         "c5": {"dpp_share_of_valu": 0.13, "mfma_pipe_cycles_each": 16, "waves_per_simd": 2},
         "c4": {"dpp_share_of_valu": 0.13, "mfma_pipe_cycles_each": 16, "waves_per_simd": 1.5}}
try:
    build = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:  # noqa: BLE001
    build = "unknown"
build = "round-6 build, measured at commit " + build

for name in ([] if ONLY else os.listdir(SRC)):
    if name.startswith("bench_") and name.endswith(".json") and os.path.getsize(os.path.join(SRC, name)) > 10:
        shutil.copy(os.path.join(SRC, name), os.path.join(PRO, "r6_" + name))
    if name.startswith("kernel_stats_"):
        shutil.copy(os.path.join(SRC, name), os.path.join(PRO, "r6_fused_" + name.replace("kernel_stats_", "").replace(".csv", "") + "_kernel_stats.csv"))
pmc = json.load(open(os.path.join(PRO, "r6_fused_pmc.json"))) if ONLY else {}
traffic = json.load(open(os.path.join(PRO, "hbm_traffic.json")))
for tag, key in KEYS.items():
    if ONLY and tag not in ONLY:
        continue
    p = os.path.join(SRC, "pmc_%s.json" % tag)
    if not os.path.exists(p):
        print("missing", p, file=sys.stderr)
        continue
    d = json.load(open(p))
    pmc[tag] = d
    fwd = {k.split("(")[0].strip(): v for k, v in d.items() if any(f in k for f in FORWARD)}
    if not fwd or any("FETCH_SIZE" not in v or "WRITE_SIZE" not in v for v in fwd.values()):
        print("incomplete FETCH/WRITE passes for", tag, "- traffic entry left as it is", file=sys.stderr)
        continue
    total = int(sum((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) for v in fwd.values()) * 1024)
    kernels = None
    bl = os.path.join(SRC, "bench_%s.json" % tag.replace("k10b", "k10"))
    if os.path.exists(bl) and os.path.getsize(bl) > 10:
        kernels = json.loads(open(bl).read().strip().split("\n")[-1])["roofline"]["kernel"]
    entry = {"FETCH_SIZE_KiB": {k: v["FETCH_SIZE"] for k, v in fwd.items()}, "WRITE_SIZE_KiB": {k: v["WRITE_SIZE"] for k, v in fwd.items()},
             "hbm_bytes_per_forward": total, "kernel": build, "kernels": kernels,
             "source": "tools/measure_round6.sh -> tools/pmc3.sh passes 3 and 4 (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 tools/run_forward.py), mean per dispatch"}
    if all("SQ_INSTS_VALU" in v for v in fwd.values()):
        entry["valu_insts_per_forward"] = sum(v["SQ_INSTS_VALU"] for v in fwd.values())
        entry["mfma_insts_per_forward"] = sum(v.get("SQ_INSTS_MFMA", 0.0) for v in fwd.values())
        entry.update(MODEL.get(tag, {}))
    traffic[key] = entry
    print(key, total, kernels)
json.dump(pmc, open(os.path.join(PRO, "r6_fused_pmc.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(PRO, "hbm_traffic.json"), "w"), indent=1)
