#!/usr/bin/env python
"""Per-CU timeline of the 256-row GEMM kernel (gemm_bf16_p8_kernel): where a tile's time goes BETWEEN the K loops.

Builds / loads lib/libdemovlp_hip_stamp.so (gemm.hip under -DDVLP_STAMP: every workgroup records entry, first data landed,
K loop done, stores issued, stores acknowledged on the chip-wide 100 MHz s_memrealtime clock + its HW_ID / XCC_ID), runs the
K = 768 products of one ViT layer and prints, per shape: kernel span, per-phase medians, and -- per CU -- the gap between one
workgroup's last stamp and the next workgroup's entry (launch gap), i.e. what a persistent kernel could overlap.

    python tools/p8_timeline.py            (on the GPU box; `python -m demovlp_amd.build --stamp` first, here or there)
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DEMOVLP_HIP_LIB"] = os.path.join(ROOT, "demovlp_amd", "lib", "libdemovlp_hip_stamp.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402

if not os.path.exists(os.environ["DEMOVLP_HIP_LIB"]):
    from demovlp_amd.build import build_stamp
    build_stamp()
from demovlp_amd import _lib  # noqa: E402

lib = _lib.load()
lib.dvlp_p8_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.dvlp_p8_stamp_buffer.restype = ctypes.c_int
dev = "cuda"
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
g = torch.Generator(device=dev).manual_seed(0)
M0 = 18496
if os.environ.get("P8_ALIAS"):              # every workgroup loads tile (0, 0)'s operands: the memory system taken out of the K loop
    lib.dvlp_p8_alias.argtypes = [ctypes.c_int]
    lib.dvlp_p8_alias(1)
    print("# P8_ALIAS=1: all workgroups load the operands of tile (0, 0)")
lib.dvlp_dev_gemm_p8_mode(2)                    # also for grids the dispatch would give to the 128-row kernel
lib.dvlp_dev_gemm_p8_persistent(0)              # the stamps live in the one-tile-per-workgroup kernel
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [("qkv fwd (bias)", M0, 2304, 768, 0, 0, "b"), ("proj fwd (bias+res)", M0, 768, 768, 0, 0, "br"), ("fc1 fwd (gelu, aux out)", M0, 3072, 768, 0, 1, "ba"),
          ("fc2 dX (gelu', aux in)", M0, 3072, 768, 1, 2, "a"), ("fc2 fwd (bias+res) K=3072", M0, 768, 3072, 0, 0, "br"),
          # is the 5-10 us epilogue a per-CU limit or the chip's store rate?  the same tiles on a quarter / half of the CUs
          ("fc1 fwd on 60 CUs", 5 * 256, 3072, 768, 0, 1, "ba"), ("fc1 fwd on 132 CUs", 11 * 256, 3072, 768, 0, 1, "ba"), ("fc1 fwd on 252 CUs", 21 * 256, 3072, 768, 0, 1, "ba")]
if len(sys.argv) > 1:
    shapes = [s_ for s_ in shapes if any(k in s_[0] for k in sys.argv[1:])]
for label, M, N, K, tb, flags, ops_ in shapes:
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    B = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16() if not tb else (torch.randn(K, N, device=dev, generator=g) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(N, device=dev) if "b" in ops_ else None
    res = torch.randn(M, N, device=dev, generator=g).bfloat16() if "r" in ops_ else None
    aux = torch.randn(M, N, device=dev, generator=g).bfloat16() if "a" in ops_ else None
    ntile = ((M + 255) // 256) * ((N + 255) // 256)
    buf = torch.zeros(ntile * 48, device=dev, dtype=torch.int64)

    def run():
        rc = lib.dvlp_gemm(1, 0, tb, M, N, K, P(A), K, P(B), N if tb else K, P(C), N, P(bias), P(res), N if res is not None else 0, P(aux),
                           N if aux is not None else 0, flags, 1.0, st)
        assert rc == 0, rc
    lib.dvlp_p8_stamp_buffer(ctypes.c_void_p(0))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 50
    lib.dvlp_p8_stamp_buffer(P(buf))
    run()
    torch.cuda.synchronize()
    lib.dvlp_p8_stamp_buffer(ctypes.c_void_p(0))
    full = buf.cpu().numpy().reshape(ntile, 2, 24)
    r = full[:, 0, :8]
    if not (r[:, 0] == np.arange(ntile)).all():
        print(f"\n=== {label}: not run on the 256-row kernel (no stamps) -- skipped")
        continue
    t = (r[:, 3:8] - r[:, 3].min()) * 0.01                     # us
    cu = ((r[:, 2] & 0xF) << 8) | ((r[:, 1] >> 8) & 0xFF)      # (xcc, se, sh, cu)
    ncu = len(np.unique(cu))
    span = t[:, 4].max()
    print(f"\n=== {label}: M={M} N={N} K={K}: {us:.1f} us/launch ({2.0 * M * N * K / us / 1e6:.0f} TF), {ntile} tiles on {ncu} CUs, stamped span {span:.1f} us")
    ph = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 4] - t[:, 0]], 1)
    names = ("entry->first data", "K loop", "epilogue (stores issued)", "store drain (vmcnt 0)", "whole workgroup")
    for i, n in enumerate(names):
        print(f"    {n:26s} median {np.median(ph[:, i]):6.2f}  p10 {np.percentile(ph[:, i], 10):6.2f}  p90 {np.percentile(ph[:, i], 90):6.2f} us")
    gaps, gaps3, rounds = [], [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(t[idx, 0])]
        rounds.append(len(idx))
        for p_, n_ in zip(idx[:-1], idx[1:]):
            gaps.append(t[n_, 0] - t[p_, 4])
            gaps3.append(t[n_, 0] - t[p_, 3])
    gaps, gaps3 = np.array(gaps), np.array(gaps3)
    print(f"    workgroups per CU: min {min(rounds)} max {max(rounds)};  first-round entry skew p90 {np.percentile(np.sort(t[:, 0])[:ncu], 90):.2f} us")
    if len(gaps):
        print(f"    next entry - stores acknowledged: median {np.median(gaps):6.2f}  p10 {np.percentile(gaps, 10):6.2f}  p90 {np.percentile(gaps, 90):6.2f} us")
        print(f"    next entry - stores issued      : median {np.median(gaps3):6.2f}  p10 {np.percentile(gaps3, 10):6.2f}  p90 {np.percentile(gaps3, 90):6.2f} us")
    busy = (t[:, 2] - t[:, 1]).sum() / (ncu * span)
    print(f"    share of CU-time inside K loops: {busy:.3f}   (K-loop-only time would be {np.median(ph[:, 1]) * max(rounds):.1f} us)")
    # inside the K loop: shader-clock cycles per phase kind q (0: reads B_lo + A_lo, 1: reads B_hi, 2: reads A_hi, 3: no reads), summed over the tile's K tiles
    nkt = K // 64
    for grp in (0, 1):
        ph = full[:, grp, 8:24].reshape(ntile, 4, 4).astype(np.float64) / nkt
        med = np.median(ph, axis=0)
        tot = med.sum()
        print(f"    K loop, wave group {grp} (cycles per K tile, median over tiles; total {tot:.0f}):")
        for q in range(4):
            print(f"        phase q{q}: load segment {med[q, 0]:6.0f}   first barrier {med[q, 1]:6.0f}   MFMA cluster (incl. fragment landing) {med[q, 2]:6.0f}   second barrier {med[q, 3]:6.0f}")
    last = np.sort(t[:, 4])
    print(f"    tail: 50 % of tiles done at {last[len(last) // 2]:.1f} us, 90 % at {last[int(len(last) * 0.9)]:.1f}, all at {last[-1]:.1f}")
