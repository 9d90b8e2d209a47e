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


import subprocess
TRAFFIC_ONLY = os.environ.get("DVLP_TRAFFIC_ONLY") is not None      # on the GPU box, between the PMC passes and the bench line (tools/profile_round.sh)
try:
    GIT_HEAD = os.environ.get("DVLP_GIT_HEAD") or subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    if not os.environ.get("DVLP_GIT_HEAD") and subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "demovlp_amd/csrc"], capture_output=True, text=True).stdout.strip():
        GIT_HEAD = (GIT_HEAD or "?") + "+uncommitted-csrc"
except Exception:
    GIT_HEAD = None
try:
    CSRC_SHA = open(os.path.join(SRC, "csrc_sha.txt")).read().strip()         # hashed on the GPU box, of the tree that was profiled
except Exception:
    CSRC_SHA = None
PROV = {"git_head": GIT_HEAD, "csrc_sha": CSRC_SHA}
line = ""
if not TRAFFIC_ONLY:
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
           **PROV, "gemm_launches": fn, "fetch_size_kib_per_launch": fs / fn, "write_size_kib_per_launch": ws / wn,
           "gemm_family_traffic_bytes_per_launch": traffic}
    json.dump(out, open(os.path.join(P, tag + "_gemm_hbm_traffic_pmc.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
if TRAFFIC_ONLY:
    sys.exit(0)
# MFMA-busy, LDS bank conflicts and parked-wave share per kernel family (third PMC pass)
mc = one("mfma/**/*counter_collection.csv")
if mc:
    fam = lambda n: ("gemm_256x256 (p8)" if "p8_kernel" in n else "gemm_wgrad_group (p8g)" if "p8_group_kernel" in n else
                     "gemm_128x128 (glds)" if "glds_kernel" in n else "attention_space (sattn, round 5)" if "sattn" in n else "attention_mfma" if "mattn" in n else
                     "local_loss_softmax" if "xsoftmax" in n else "layernorm" if "ln_" in n else None)
    agg, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
    for r in csv.DictReader(open(mc)):
        f = fam(r["Kernel_Name"])
        if f:
            agg[f][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[f] += 1
    out = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES "
                     "SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 (tools/profile_round.sh)",
           "note": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the gfx94x MfmaUtil formula with GRBM_GUI_ACTIVE, which "
                   "rocprofv3 reports summed over the 8 XCDs, brought back to one clock (cross-check: p8 kernels 0.30 busy ~ 750 TFLOP/s of 2500; ROCm 7.2 ships no gfx950 "
                   "derived-counter section); lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES "
                   "(share of wave time at s_waitcnt / barriers); mfma_ops_bf16 are 512-FLOP units per MI355X_MICROARCH.md",
           **PROV, "families": {}}
    for f, c in agg.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        out["families"][f] = {"launches": cnt[f],
                              "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(gui / 8.0 * 256 * 4, 1.0), 4),
                              "lds_conflict": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4),
                              "parked": round(c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0), 4),
                              "gpu_cycles_per_launch": round(gui / 8.0 / max(cnt[f], 1)),
                              "mfma_mops_bf16_per_launch": round(c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) / max(cnt[f], 1))}
    json.dump(out, open(os.path.join(P, tag + "_mfma_lds_pmc.json"), "w"), indent=1)
    print(json.dumps(out["families"], indent=1))
# L2<->fabric traffic per kernel (same two PMC passes): which launches re-read their operands
if fc and wc:
    import re

    def by_kernel(path, counter):
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                k = re.sub(r"\(.*", "", r["Kernel_Name"])[:90]
                agg[k] += float(r["Counter_Value"])
                cnt[k] += 1
        return agg, cnt
    fa, fn_ = by_kernel(fc, "FETCH_SIZE")
    wa, _ = by_kernel(wc, "WRITE_SIZE")
    rows = sorted(((2.0 * fa[k] + wa.get(k, 0.0)) * 1024.0, k) for k in fa)[::-1]
    total = sum(t for t, _ in rows)
    table = [{"kernel": k, "launches": fn_[k], "read_mb_per_launch": round(2.0 * fa[k] * 1024 / fn_[k] / 1e6, 1),
              "write_mb_per_launch": round(wa.get(k, 0.0) * 1024 / fn_[k] / 1e6, 1), "share_of_all_traffic": round(t / total, 4)} for t, k in rows[:25]]
    json.dump({"source": "the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh (KiB counters; reads doubled per MI355X_MICROARCH.md); L2<->fabric, "
                         "Infinity-Cache hits included, so HBM traffic proper is lower for operands that stay in the 256 MB MALL",
               **PROV, "kernels": table}, open(os.path.join(P, tag + "_traffic_by_kernel.json"), "w"), indent=1)

f32 = os.path.join(SRC, "bench_f32_b16.json")
if os.path.exists(f32):
    l32 = [l for l in open(f32) if l.startswith("{")]
    if l32:
        open(os.path.join(P, tag + "_bench_f32_b16.json"), "w").write(l32[-1])
for jf in ("bench_f8_r30_b32.json", "ln_fusion_bound.json"):        # config 4's shape (B = 32, F = 8, R = 30); the timing-only LayerNorm-forward ablation
    src = os.path.join(SRC, jf)
    if os.path.exists(src):
        lj = [l for l in open(src) if l.startswith("{")]
        if lj:
            open(os.path.join(P, tag + "_" + jf), "w").write(lj[-1])
for extra in ("tile_sweep.txt", "loss_gemm_bench.txt", "select_bench.txt", "select_bench_f32.txt", "xfused_check.txt", "xloss_bench.txt", "epi_bench.txt", "lib_gemm_ref.txt", "gemm_shapes.txt",
              "input_bench.txt", "eval_bench.txt", "tile_height_bench.txt", "attn_bench.txt", "loss_head_bench.txt", "step_timeline.txt"):
    src = os.path.join(SRC, extra)
    if os.path.exists(src):
        keep = [l for l in open(src) if "amdgpu.ids" not in l]
        open(os.path.join(P, tag + "_" + extra), "w").writelines(keep)
print(line[:400])
