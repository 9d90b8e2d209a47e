#!/usr/bin/env python
"""256-row against 224-row tiles of the 256 x 256 ping-pong GEMM kernel on the object tower's forward / dX shapes (M = 18496 tokens):
interleaved timings in one process.      python tools/tile_height_bench.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from demovlp_amd import _lib  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

lib = _lib.load()
dev = "cuda"
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
g = torch.Generator(device=dev).manual_seed(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("TH_M", "18496"))
shapes = [("qkv fwd (bias)", 2304, 768, 0, 0, "b"), ("proj fwd (bias+res)", 768, 768, 0, 0, "br"), ("fc1 fwd (gelu, aux out)", 3072, 768, 0, 1, "ba"),
          ("fc2 fwd (bias+res)", 768, 3072, 0, 0, "br"), ("fc2 dX (gelu', aux in)", 3072, 768, 1, 2, "a"), ("fc1 dX", 768, 3072, 1, 0, ""),
          ("proj dX", 768, 768, 1, 0, ""), ("qkv dX", 768, 2304, 1, 0, "")]
tot = {0: 0.0, 1: 0.0}
for label, N, K, tb, flags, ops_ in shapes:
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    B = ((torch.randn(N, K, device=dev, generator=g) if not tb else torch.randn(K, N, device=dev, generator=g)) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g) if "b" in ops_ else None
    res = torch.randn(M, N, device=dev, generator=g).bfloat16() if "r" in ops_ else None
    aux = torch.randn(M, N, device=dev, generator=g).bfloat16() if "a" in ops_ else None

    def run():
        rc = lib.dvlp_gemm(1, 0, tb, M, N, K, P(A), K, P(B), N if tb else K, P(C), N, P(bias), P(res), N if res is not None else 0, P(aux),
                           N if aux is not None else 0, flags, 1.0, st)
        assert rc == 0, rc
    times = {0: [], 1: []}
    for r in range(7):
        for mode in (0, 1):
            lib.dvlp_dev_gemm_p8_short_tiles(mode)
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) * 100)
    fl = 2.0 * M * N * K
    med = {m: sorted(t)[len(t) // 2] for m, t in times.items()}
    for m in (0, 1):
        tot[m] += med[m]
    print(f"{label:26s} N={N:5d} K={K:5d}   256-row tiles {med[0]:7.1f} us ({fl / med[0] / 1e6:5.0f} TF)   automatic height {med[1]:7.1f} us ({fl / med[1] / 1e6:5.0f} TF)   {med[0] / med[1]:.3f}x")
print(f"sum over one layer's forward + dX products: {tot[0]:.1f} -> {tot[1]:.1f} us ({tot[0] / tot[1]:.3f}x)")
lib.dvlp_dev_gemm_p8_short_tiles(1)
