#!/usr/bin/env python
"""Instruction histogram of one kernel in a hipcc -S listing:  python tools/isa_hist.py file.s mangled_name [top]"""
import collections
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ":")
j = s.index("s_endpgm", i)
cnt = collections.Counter()
cat = collections.Counter()
for l in s[i:j].split("\n"):
    l = l.strip()
    if not l or l.startswith((".", ";", "_")) or l.endswith(":"):
        continue
    op = l.split()[0]
    cnt[op] += 1
    cat["mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
        "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "salu"] += 1
print(sum(cnt.values()), dict(cat))
for k, v in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
    print("%5d %s" % (v, k))
