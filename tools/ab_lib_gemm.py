#!/usr/bin/env python
"""A/B of two builds of the library on the object tower's forward / dX products in ONE process (interleaved rounds):
    python tools/ab_lib_gemm.py demovlp_amd/lib/libdemovlp_hip_r3.so demovlp_amd/lib/libdemovlp_hip.so
(the reference build comes from tools/build_ref_lib.sh <rev> <name>).  Also reports whether the outputs are bit-equal."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from demovlp_amd import _lib  # noqa: E402

paths = sys.argv[1:3]
libs = []
for p in paths:
    lib = ctypes.CDLL(os.path.abspath(p))
    for name, (ret, argtypes) in _lib._SIGS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = ret
    libs.append(lib)
dev = "cuda"
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
g = torch.Generator(device=dev).manual_seed(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("TH_M", "18496"))
shapes = [("qkv fwd (bias)", 2304, 768, 0, 0, "b"), ("proj fwd (bias+res)", 768, 768, 0, 0, "br"), ("fc1 fwd (gelu, aux out)", 3072, 768, 0, 1, "ba"),
          ("fc2 dX (gelu', aux in)", 3072, 768, 1, 2, "a"), ("proj dX", 768, 768, 1, 0, ""),
          ("fc2 fwd (bias+res)", 768, 3072, 0, 0, "br"), ("fc1 dX", 768, 3072, 1, 0, ""), ("qkv dX", 768, 2304, 1, 0, "")]
tot = [0.0, 0.0]
for label, N, K, tb, flags, ops_ in shapes:
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    B = ((torch.randn(N, K, device=dev, generator=g) if not tb else torch.randn(K, N, device=dev, generator=g)) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev, generator=g) if "b" in ops_ else None
    res = torch.randn(M, N, device=dev, generator=g).bfloat16() if "r" in ops_ else None
    aux0 = torch.randn(M, N, device=dev, generator=g).bfloat16() if "a" in ops_ else None

    def run(lib, C, aux):
        rc = lib.dvlp_gemm(1, 0, tb, M, N, K, P(A), K, P(B), N if tb else K, P(C), N, P(bias), P(res), N if res is not None else 0, P(aux),
                           N if aux is not None else 0, flags, 1.0, st)
        assert rc == 0, rc
    outs = []
    for lib in libs:
        C = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        aux = aux0.clone() if aux0 is not None else None
        run(lib, C, aux)
        torch.cuda.synchronize()
        outs.append((C, aux))
    eq = torch.equal(outs[0][0], outs[1][0]) and (aux0 is None or torch.equal(outs[0][1], outs[1][1]))
    maxd = (outs[0][0].float() - outs[1][0].float()).abs().max().item()
    times = [[], []]
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for r in range(7):
        for i, lib in enumerate(libs):
            for _ in range(2):
                run(lib, C, aux0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(lib, C, aux0)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) * 100)
    fl = 2.0 * M * N * K
    med = [sorted(t)[len(t) // 2] for t in times]
    tot[0] += med[0]; tot[1] += med[1]
    print(f"{label:26s} N={N:5d} K={K:5d}   A {med[0]:7.1f} us ({fl / med[0] / 1e6:5.0f} TF)   B {med[1]:7.1f} us ({fl / med[1] / 1e6:5.0f} TF)   A/B {med[0] / med[1]:.3f}x   bit-equal {eq} (max diff {maxd:.3g})", flush=True)
print(f"sum: A {tot[0]:.1f} us   B {tot[1]:.1f} us   {tot[0] / tot[1]:.3f}x")
