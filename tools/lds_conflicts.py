#!/usr/bin/env python3
"""Bank-conflict audit of the structured-tile kernel's LDS accesses (mirrors the address maths of cheb_struct_kernel.h).
ds_read_b128: 4 groups of 16 lanes {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32), bank (a/4)%64; a group costs as many
cycles as the busiest bank has distinct addresses.  ds_write_b128: 8 groups of 8 consecutive lanes, bank (a/4)%32."""
import re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
S, P2 = 24, 13
HP = S * P2
PLANE = 2 * HP * 64
_src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deepsphere-cosmo-tf2_amd", "csrc", "cheb_struct_tables.h")).read()
tab = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{2})", _src[_src.index("kStructBlock[128]"):])][:128]
RG = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
RG = RG + [[l + 32 for l in g] for g in RG]
def cell_off(gx, gy): return ((gx & 1) * HP + gy * P2 + (gx >> 1)) * 64
def cell_f(gx, gy): return (gx & 1) | ((gy & 1) << 1)
def read_cost(addrs):  # addrs: 64 byte addresses (None = inactive)
    tot = 0
    for g in RG:
        banks = {}
        for l in g:
            a = addrs[l]
            if a is None: continue
            for d in range(4):
                banks.setdefault(((a // 4) + d) % 64, set()).add(a)
        tot += max([len(v) for v in banks.values()] or [1])
    return tot
def write_cost(addrs):
    tot = 0
    for g in range(8):
        banks = {}
        for l in range(8 * g, 8 * g + 8):
            a = addrs[l]
            if a is None: continue
            for d in range(4):
                banks.setdefault(((a // 4) + d) % 32, set()).add(a)
        tot += max([len(v) for v in banks.values()] or [1])
    return tot
tot_r = tot_w = n_r = n_w = 0
for wave in range(8):
    # gather window reads and own-cell writes
    for wy in range(4):
        for wx in range(4):
            addrs = []
            for lane in range(64):
                blk = tab[wave * 16 + lane // 4]; q = lane % 4
                if blk == 0xff: addrs.append(None); continue
                bx, by = blk & 15, blk >> 4
                gx, gy = 2 * bx + wx, 2 * by + wy
                addrs.append(cell_off(gx, gy) + 16 * (q ^ cell_f(gx, gy)))
            c = read_cost(addrs); tot_r += c; n_r += 1
            if c != 4: print(f"gather read wave {wave} window ({wx},{wy}): {c} cycles")
            if wx in (1, 2) and wy in (1, 2):
                c = write_cost(addrs); tot_w += c; n_w += 1
                if c != 8: print(f"gather write wave {wave} cell ({wx},{wy}): {c} cycles")
    # contraction operand reads
    for e in range(2):
        addrs = []
        for lane in range(64):
            r, h = lane & 31, lane >> 5
            gx, gy = 4 + (r & 15), 4 + 2 * wave + (r >> 4)
            addrs.append(cell_off(gx, gy) + 16 * ((2 * h + e) ^ cell_f(gx, gy)))
        c = read_cost(addrs); tot_r += c; n_r += 1
        if c != 4: print(f"contract read wave {wave} half {e}: {c} cycles")
    # store transposition block
    for tq in range(4):
        addrs = [ (l & 31) * 144 + (8 * tq + 4 * (l >> 5)) * 4 for l in range(64)]
        c = write_cost(addrs)
        if c != 8 and wave == 0: print(f"store scratch write tq {tq}: {c} cycles")
    for i in range(4):
        addrs = [((l >> 3) + 8 * i) * 144 + 16 * (l & 7) for l in range(64)]
        c = read_cost(addrs)
        if c != 4 and wave == 0: print(f"store scratch read round {i}: {c} cycles")
print(f"reads: {tot_r} cycles for {n_r} wave-instructions (ideal {4 * n_r}); writes: {tot_w} for {n_w} (ideal {8 * n_w})")
