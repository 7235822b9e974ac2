#!/usr/bin/env python3
"""Opcode-class histogram of a kernel from the compiler's assembly (hipcc --offload-device-only -S), whole kernel and per
loop body: which instructions the shipped cheb_qstrip5_kernel issues per step, by class.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ideepsphere-cosmo-tf2_amd/csrc -fno-slp-vectorize \
          --offload-device-only -S deepsphere-cosmo-tf2_amd/csrc/cheb_qstrip.hip -o /tmp/q.s
    python3 tools/opcode_hist.py /tmp/q.s _ZN4dsph19cheb_qstrip5_kernelILb1ELb0EEEvNS_10QStripArgsE

A loop = the lines between a label and the LAST backward branch to it; the innermost loops of the quad-strip kernel are the H
role's (288 matrix instructions) and the L role's (72, and the y stores) three-step bodies (the step is unrolled three times)."""
import collections
import re
import sys


def classify(op, line):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        if "dpp" in op or "row_sh" in line or "quad_perm" in line or "row_bcast" in line:
            return "valu_dpp"
        if op.startswith(("v_fmac_f32", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_pk_fma", "v_pk_mul", "v_pk_add", "v_max_f32", "v_min_f32")):
            return "valu_fp32"
        if op.startswith(("v_cvt", "v_pack", "v_perm", "v_and", "v_or", "v_lshl", "v_lshr", "v_bfe", "v_bfi", "v_xor", "v_alignbit")):
            return "valu_cvt_bit"
        if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
            return "valu_mov"
        if op.startswith(("v_cmp", "v_cndmask")):
            return "valu_cmp_sel"
        if op.startswith(("v_add_u32", "v_add_co", "v_addc", "v_sub_u32", "v_mad_u", "v_mad_i", "v_mul_lo", "v_mul_hi", "v_add3", "v_lshl_add", "v_lshl_or", "v_add_lshl", "v_mad_u64", "v_ashr", "v_subrev", "v_min_i", "v_max_i", "v_min_u", "v_max_u", "v_med3", "v_readfirstlane", "v_readlane", "v_mbcnt")):
            return "valu_int_addr"
        return "valu_other"
    if op.startswith("ds_"):
        return "lds_read" if ("read" in op or "load" in op) else "lds_write"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
        return "vmem_store"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "s_branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    insts, labels = [], {}
    for ln in body:
        t = ln.split(";")[0].strip()
        if not t:
            continue
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):$", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if t.startswith("."):
            continue
        insts.append((t.split()[0], t))
    hist = collections.Counter(classify(op, t) for op, t in insts)
    print(f"kernel {kernel}: {len(insts)} instructions")
    order = ["valu_fp32", "valu_dpp", "valu_cvt_bit", "valu_mov", "valu_cmp_sel", "valu_int_addr", "valu_other", "mfma", "lds_read", "lds_write",
             "vmem_load", "vmem_store", "salu", "smem", "s_waitcnt", "s_nop", "s_barrier", "s_branch", "other"]

    def show(h, title):
        tot_valu = sum(v for k, v in h.items() if k.startswith("valu"))
        print(f"  {title}: VALU {tot_valu} (" + ", ".join(f"{k[5:]} {h[k]}" for k in order if k.startswith("valu") and h[k]) + ")")
        print("    " + ", ".join(f"{k} {h[k]}" for k in order if not k.startswith("valu") and h[k]))
    show(hist, "whole kernel")
    # loops: backward branches
    loops = {}
    for i, (op, t) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops[tgt] = max(loops.get(tgt, 0), i)
    # the step bodies: loops of at least 1,000 instructions that hold no other such loop (the kernel's innermost loops are three
    # steps each; with uniform branches inside a step the compiler makes several back edges of one loop: variants are listed once
    # per distinct size class)
    big = sorted((e - labels[l] + 1, l, labels[l], e) for l, e in loops.items() if e - labels[l] + 1 >= 1000)
    inner = [b for b in big if not any(o is not b and o[2] >= b[2] and o[3] <= b[3] and (o[2], o[3]) != (b[2], b[3]) for o in big)]
    seen = set()
    for n, l, a, e in inner:
        h = collections.Counter(classify(op, t) for op, t in insts[a:e + 1])
        key = (h["mfma"], h["vmem_store"], n // 64)
        if key in seen:
            continue
        seen.add(key)
        show(h, f"innermost loop {l}: {n} instructions")
        if "-v" in sys.argv:
            other = collections.Counter(op for op, t in insts[a:e + 1] if classify(op, t) in ("valu_other", "valu_mov", "valu_cvt_bit", "valu_int_addr", "valu_cmp_sel"))
            print("    non-stencil vector opcodes: " + ", ".join(f"{k} {v}" for k, v in other.most_common(30)))


if __name__ == "__main__":
    main()
