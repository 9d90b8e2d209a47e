#!/usr/bin/env python
"""Condensed view of one kernel's assembly: runs of scratch traffic, MFMAs, barriers, waits, branches and labels, in order.
    python tools/asm_regions.py file.s mangled-name-substring"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(tuple(":" + x for x in ("",))) or (l.startswith("_Z") and key in l and ": " in l))
out, prev, cnt, first = [], None, 0, 0
def cls(l):
    t = l.strip().split()
    if not t: return None
    op = t[0]
    if op.startswith("scratch_") or (op.startswith("buffer_") and "offen" in l and "s[0:3]" in l): return "SCRATCH_" + ("ST" if "store" in op else "LD")
    if op.startswith("v_mfma"): return "mfma"
    if op == "s_barrier": return "BARRIER"
    if op == "s_waitcnt": return "wait " + " ".join(t[1:])
    if op.startswith("s_cbranch") or op == "s_branch": return op + " " + t[1]
    if op.startswith("global_load_lds") or (op.startswith("global_load") and "lds" in l): return "glds"
    if op.startswith("global_store") or op.startswith("global_load"): return op.split("_")[1] + "_global"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if re.match(r"^\.LBB\d+_\d+:", l.strip()): return l.strip()
    if op == "s_endpgm": return "END"
    if op.startswith("v_readlane") or op.startswith("v_writelane"): return "sgpr_spill_lane"
    return None
n_other = 0
for i in range(start, len(lines)):
    c = cls(lines[i])
    if c is None:
        if lines[i].strip() and not lines[i].strip().startswith((";", ".")): n_other += 1
        continue
    if c == prev: cnt += 1
    else:
        if prev is not None: out.append((first, prev, cnt, n_prev_other))
        prev, cnt, first, n_prev_other = c, 1, i + 1, n_other
        n_other = 0
    if c == "END": break
out.append((first, prev, cnt, 0))
for f, p, c, o in out:
    print(f"{f:7d}  (+{o:3d} other)  {p}" + (f"  x{c}" if c > 1 else ""))
