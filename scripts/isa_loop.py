#!/usr/bin/env python3
"""Compressed view of a kernel's hottest loop from a hipcc -S dump: one letter per instruction
(M mfma, v VALU, t transcendental, r ds_read, w ds_write, g global/buffer load, s store, L lds-dma, W s_waitcnt, B barrier, . other scalar).
    python scripts/isa_loop.py file.s '<mangled-name-substring>'"""
import re, sys
txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l and ":" in l.split()[0])
end = next(i for i in range(start + 1, len(txt)) if txt[i].startswith("\t.end_amdhsa_kernel") or txt[i].startswith(".Lfunc_end"))
body = txt[start:end]
# basic blocks
blocks, cur, name = [], [], "entry"
for l in body:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1)
    elif l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
        cur.append(l.strip())
blocks.append((name, cur))
def cls(i):
    op = i.split()[0]
    if "mfma" in op: return "M"
    if op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_log") or op.startswith("v_sqrt"): return "t"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "r"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "w"
    if "load" in op and "lds" in i: return "L"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "g"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "s"
    if op.startswith("s_waitcnt"): return "W"
    if op.startswith("s_barrier"): return "B"
    if op.startswith("v_"): return "v"
    return "."
best = max(blocks, key=lambda b: sum(1 for i in b[1] if "mfma" in i))
print("kernel", txt[start][:90]); print("hot block", best[0], "instructions", len(best[1]))
seq = "".join(cls(i) for i in best[1])
for k in range(0, len(seq), 150): print(seq[k:k + 150])
print({c: seq.count(c) for c in "MvtrwgsLWB."})
for i in best[1]:
    if i.startswith("s_waitcnt") and "vmcnt" in i: print("   ", i, "at", best[1].index(i))
if len(sys.argv) > 3:
    print("--- all blocks")
    for n, b in blocks:
        sq = "".join(cls(i) for i in b)
        if len(sq) > 8: print(n, len(sq), {c: sq.count(c) for c in "MvtrwgsLWB" if sq.count(c)})
