#!/usr/bin/env python
"""Persistent vs one-tile-per-workgroup form of the 256-row GEMM kernel (224-row tiles) on multi-round outputs: agreement of the
outputs (the persistent form starts its accumulators from the bias, so fp32 sums differ in the last bit before the bf16 rounding),
a repeat screen for the cross-tile prefetch / counted-vmcnt logic, and interleaved timings.      python tools/p8p_bench.py"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from demovlp_amd import _lib  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--screen", type=int, default=30)
a = ap.parse_args()
lib = _lib.load()
dev = "cuda"
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
g = torch.Generator(device=dev).manual_seed(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M0 = 18496
shapes = [("qkv fwd (bias)", M0, 2304, 768, 0, 0, "b"), ("fc1 fwd (gelu, aux out)", M0, 3072, 768, 0, 1, "ba"), ("plain N=2304", M0, 2304, 768, 0, 0, ""),
          ("qkv dX-like KR K=2304 x4 rows", 4 * M0, 768, 2304, 1, 0, ""), ("text qkv M=6400", 6400, 2304, 768, 0, 0, "b"), ("text fc1 (gelu) M=25600", 25600, 3072, 768, 0, 1, "ba")]
for label, M, N, K, tb, flags, ops_ in shapes:
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    B = ((torch.randn(N, K, device=dev, generator=g) if not tb else torch.randn(K, N, device=dev, generator=g)) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev, generator=g) if "b" in ops_ else None

    def run(C, aux):
        rc = lib.dvlp_gemm(1, 0, tb, M, N, K, P(A), K, P(B), N if tb else K, P(C), N, P(bias), None, 0, P(aux), N if aux is not None else 0, flags, 1.0, st)
        assert rc == 0, rc

    def fresh():
        C = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        aux = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16) if "a" in ops_ else None
        return C, aux
    lib.dvlp_dev_gemm_p8_persistent(0)
    C0, aux0 = fresh()
    run(C0, aux0)
    torch.cuda.synchronize()
    assert not torch.isnan(C0.float()).any()
    lib.dvlp_dev_gemm_p8_persistent(1)
    C1, aux1 = fresh()
    run(C1, aux1)
    torch.cuda.synchronize()
    d = (C1.float() - C0.float()).abs()
    ulp = C0.float().abs() * 2.0 ** -7 + 1e-30
    print(f"\n=== {label}: M={M} N={N} K={K}: persistent vs one-tile: identical {float((d == 0).float().mean()) * 100:.3f} % of elements, max |diff| / bf16 ulp {float((d / ulp).max()):.2f}, NaNs {int(torch.isnan(C1.float()).sum())}"
          + (f"; aux identical {float((aux1 == aux0).float().mean()) * 100:.3f} %" if aux0 is not None else ""))
    bad = 0
    for _ in range(a.screen):
        C2, aux2 = fresh()
        run(C2, aux2)
        bad += int(not torch.equal(C2, C1)) + int(aux2 is not None and not torch.equal(aux2, aux1))
    print(f"    persistent form, {a.screen} repeats: {bad} differ from the first run")
    times = {0: [], 1: []}
    C, aux = fresh()
    for r in range(a.rounds):
        for mode in (0, 1):
            lib.dvlp_dev_gemm_p8_persistent(mode)
            for _ in range(2):
                run(C, aux)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(C, aux)
            e1.record()
            torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) * 100)
    fl = 2.0 * M * N * K
    med = {m: sorted(t)[len(t) // 2] for m, t in times.items()}
    print(f"    one tile / WG {med[0]:7.1f} us ({fl / med[0] / 1e6:5.0f} TF)   persistent {med[1]:7.1f} us ({fl / med[1] / 1e6:5.0f} TF)   {med[0] / med[1]:.3f}x")
lib.dvlp_dev_gemm_p8_persistent(0)
