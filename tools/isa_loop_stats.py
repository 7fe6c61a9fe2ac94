"""Instruction mix of the hot loops of a gfx950 assembly file (hipcc -save-temps): for every kernel, the innermost loops that hold
MFMAs - counts of MFMA / LDS / VMEM / VALU / accvgpr moves / waits / scratch per trip.  Usage: isa_loop_stats.py file.s"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
kern = None
blocks = []          # (kernel, label, start, end)
cur = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", l)
    if m and not l.startswith(".L"):
        kern = m.group(1)
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        if cur:
            blocks.append((cur[0], cur[1], cur[2], i))
        cur = (kern, m.group(1), i)
    if l.strip().startswith("s_endpgm") and cur:
        blocks.append((cur[0], cur[1], cur[2], i))
        cur = None
pat = {"mfma": r"v_mfma", "asm_mfma?": r";\s*inlineasm|^\s*;APP", "ds_read": r"ds_read", "ds_bpermute": r"ds_bpermute", "ds_write": r"ds_write",
       "vmem_ld": r"buffer_load_dwordx4 v|global_load", "dma": r"buffer_load.* lds", "vmem_st": r"buffer_store|global_store",
       "accvgpr": r"v_accvgpr", "waitcnt": r"s_waitcnt", "nop": r"s_nop", "scratch": r"scratch_", "barrier": r"s_barrier",
       "fma_mix": r"v_fma_mix", "valu": r"^\s*v_(?!mfma|accvgpr)"}
for k, lab, a, b in blocks:
    body = lines[a:b]
    n = sum(1 for l in body if re.search(r"v_mfma", l))
    if n < 8:
        continue
    c = {name: sum(1 for l in body if re.search(p, l)) for name, p in pat.items()}
    print(f"{k[:60]} {lab} lines {a}-{b} ({b - a}):", " ".join(f"{n_}={v}" for n_, v in c.items() if v))
