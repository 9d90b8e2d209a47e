#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel family: mean counter value per dispatch."""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    name = r["Kernel_Name"]
    fam = "gemm_bf16_glds" if "gemm_bf16_glds" in name else ("gemm_bf16" if "gemm_bf16" in name else name[:48])
    agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(fam, r["Counter_Name"])] += 1
out = {}
for fam, d in agg.items():
    out[fam] = {c: {"sum": v, "dispatches": cnt[(fam, c)], "mean_per_dispatch": v / cnt[(fam, c)]} for c, v in d.items()}
top = sorted(out.items(), key=lambda kv: -max(x["sum"] for x in kv[1].values()))[:12]
print(json.dumps(dict(top), indent=1))
