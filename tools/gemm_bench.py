#!/usr/bin/env python
"""Micro-benchmark of dvlp_gemm on the hot path's shapes (MI355X).  Usage: python tools/gemm_bench.py [--dtype bf16|fp32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--tokens", type=int, default=18496)
    ap.add_argument("--variant", type=int, default=1, help="1 = LDS-DMA kernel, 0 = register-staged kernel")
    ap.add_argument("--splitk-target", type=int, default=768)
    ap.add_argument("--ablate", type=int, default=0, help="TIMING ONLY: 1 no LDS-DMA, 2 no fragment reads, 4 no MFMA")
    ap.add_argument("--wide", type=int, default=1, help="0 never / 1 heuristic / 2 always use the 256x128 tile")
    ap.add_argument("--p8", type=int, default=0, help="0 never / 1 heuristic / 2 always use the 256x256 ping-pong kernel")
    a = ap.parse_args()
    ops.call("dvlp_dev_gemm_p8_mode", a.p8)
    ops.call("dvlp_dev_gemm_variant", a.variant)
    ops.call("dvlp_dev_gemm_splitk_target", a.splitk_target)
    ops.call("dvlp_dev_gemm_ablate", a.ablate)
    ops.call("dvlp_dev_gemm_wide_mode", a.wide)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = "cuda"
    M = a.tokens
    g = torch.Generator(device=dev).manual_seed(0)
    rows = []
    for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
        x = torch.randn(M, K, device=dev, generator=g).to(dt)
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(dt)
        dy = torch.randn(M, N, device=dev, generator=g).to(dt)
        bias = torch.zeros(N, device=dev)
        fl = 2.0 * M * N * K
        t = bench(lambda: ops.linear_fwd(x, w, bias))
        rows.append((name + " fwd  (K,K)", M, N, K, t, fl / t / 1e12))
        t = bench(lambda: ops.linear_bwd_input(dy, w))
        rows.append((name + " dX   (K,R)", M, K, N, t, fl / t / 1e12))
        t = bench(lambda: ops.linear_bwd_weight(dy, x))
        rows.append((name + " dW   (R,R)", N, K, M, t, fl / t / 1e12))
    for r in rows:
        print("%-18s M=%6d N=%5d K=%6d  %8.1f us  %7.1f TFLOP/s" % (r[0], r[1], r[2], r[3], r[4] * 1e6, r[5]))


if __name__ == "__main__":
    main()
