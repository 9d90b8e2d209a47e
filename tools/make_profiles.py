#!/usr/bin/env python
"""Turn gpurun_out/prof_round/ (tools/profile_round.sh) into the committed summaries under profiles/:
   <tag>_bench.json               the default bench.py line
   <tag>_kernel_stats.csv         rocprofv3 --kernel-trace --stats summary of the same command
   <tag>_gemm_hbm_traffic_pmc.json  per-launch L2<->fabric traffic of the GEMM family from the FETCH_SIZE / WRITE_SIZE passes
Usage: python tools/make_profiles.py r1_final"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_round")
tag = sys.argv[1] if len(sys.argv) > 1 else "r1_final"
P = os.path.join(ROOT, "profiles")


def one(pattern):
    g = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return g[0] if g else None


line = [l for l in open(os.path.join(SRC, "bench.json")) if l.startswith("{")][-1]
open(os.path.join(P, tag + "_bench.json"), "w").write(line)
st = one("trace/**/*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(P, tag + "_bench_kernel_stats.csv"))
tl = [l for l in open(os.path.join(SRC, "trace_bench.json")) if l.startswith("{")]
if tl:
    open(os.path.join(P, tag + "_bench_under_rocprof.json"), "w").write(tl[-1])


def family(name):
    return "gemm_bf16" if ("gemm_bf16" in name or "p8_group" in name or "splitk_reduce" in name) else None


def per_dispatch(path, counter):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        f = family(r["Kernel_Name"])
        if f:
            agg[f] += float(r["Counter_Value"])
            if "splitk_reduce" not in r["Kernel_Name"] and "group_reduce" not in r["Kernel_Name"]:
                cnt[f] += 1
    return {f: (agg[f], cnt[f]) for f in agg}


fc, wc = one("fetch/**/*counter_collection.csv"), one("write/**/*counter_collection.csv")
if fc and wc:
    fe, wr = per_dispatch(fc, "FETCH_SIZE"), per_dispatch(wc, "WRITE_SIZE")
    (fs, fn), (ws, wn) = fe["gemm_bf16"], wr["gemm_bf16"]
    # FETCH_SIZE / WRITE_SIZE count KiB; gfx950 reports half the bytes of wide coalesced reads -> reads are doubled
    traffic = (2.0 * fs / fn + ws / wn) * 1024.0
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py --steps 2 "
                     "--warmup 1 --no-kernel-timing --no-cpu-baseline, MI355X (tools/profile_round.sh)",
           "note": "KiB of L2<->fabric traffic (Infinity-Cache hits included); reads doubled per MI355X_MICROARCH.md (HBM section); the split-K "
                   "slab reductions are charged to the GEMM launch they belong to",
           "gemm_launches": fn, "fetch_size_kib_per_launch": fs / fn, "write_size_kib_per_launch": ws / wn,
           "gemm_family_traffic_bytes_per_launch": traffic}
    json.dump(out, open(os.path.join(P, tag + "_gemm_hbm_traffic_pmc.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
print(line[:400])
