#!/usr/bin/env python
"""What the stock library (hipBLASLt through torch.matmul / F.linear) reaches on the hot shapes -- a yardstick for the hand-written
kernels, never part of the product path.  python tools/lib_gemm_ref.py"""
import torch

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
T = 18496


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
    return a.elapsed_time(b) / iters * 1e3


for label, M, N, K, form in (("qkv fwd", T, 2304, 768, "KK"), ("proj fwd", T, 768, 768, "KK"), ("fc1 fwd", T, 3072, 768, "KK"), ("fc2 fwd", T, 768, 3072, "KK"),
                             ("fc1 dX", T, 768, 3072, "KR"), ("fc2 dX", T, 3072, 768, "KR"), ("fc1 dW", 3072, 768, T, "RR"), ("fc2 dW", 768, 3072, T, "RR"),
                             ("text fc1 fwd", 6400, 3072, 768, "KK"), ("text fc2 fwd", 6400, 768, 3072, "KK")):
    if form == "KK":
        A = torch.randn(M, K, device=dev, generator=g).bfloat16(); B = torch.randn(N, K, device=dev, generator=g).bfloat16()
        fn = lambda: torch.matmul(A, B.t())
    elif form == "KR":
        A = torch.randn(M, K, device=dev, generator=g).bfloat16(); B = torch.randn(K, N, device=dev, generator=g).bfloat16()
        fn = lambda: torch.matmul(A, B)
    else:
        A = torch.randn(K, M, device=dev, generator=g).bfloat16(); B = torch.randn(K, N, device=dev, generator=g).bfloat16()
        fn = lambda: torch.matmul(A.t(), B)
    t = bench(fn)
    print(f"{label:14s} {form} M={M:6d} N={N:5d} K={K:6d}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:7.0f} TFLOP/s (plain product, bf16 out)")
