#!/usr/bin/env python
"""The text tower's products (M = 6400 tokens) on the three bf16 kernels of this library: 128 x 128 two-stage (g128), 256 x 128 three-stage
(wide), 256 x 256 ping-pong (p8, with its automatic K split).      python tools/text_gemm_bench.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from demovlp_amd import _lib, ops  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

lib = _lib.load()
dev = "cuda"
ops.ensure_gemm_workspace(torch.device(dev), 512)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev).manual_seed(0)
M = int(os.environ.get("TG_M", "6400"))
shapes = [("qkv fwd", 2304, 768, 0, 0), ("out fwd", 768, 768, 0, 0), ("fc1 fwd (gelu)", 3072, 768, 0, 1), ("fc2 fwd", 768, 3072, 0, 0),
          ("out dX", 768, 768, 1, 0), ("qkv dX", 768, 2304, 1, 0), ("fc2 dX (gelu')", 3072, 768, 1, 2), ("fc1 dX", 768, 3072, 1, 0)]
modes = [("g128", 0, 0), ("wide 256x128", 2, 0), ("p8 256x256", 0, 2), ("dispatch default", 0, 1)]
tot = {m[0]: 0.0 for m in modes}
for label, N, K, tb, flags in shapes:
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    B = ((torch.randn(N, K, device=dev, generator=g) if not tb else torch.randn(K, N, device=dev, generator=g)) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g) if not tb else None
    aux = torch.randn(M, N, device=dev, generator=g).bfloat16() if flags else None
    row = []
    for name, wide, p8 in modes:
        lib.dvlp_dev_gemm_wide_mode(wide)
        lib.dvlp_dev_gemm_p8_mode(p8)

        def run():
            rc = lib.dvlp_gemm(1, 0, tb, M, N, K, P(A), K, P(B), N if tb else K, P(C), N, P(bias), None, 0, P(aux), N if aux is not None else 0, flags, 1.0, st)
            assert rc == 0, rc
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        tot[name] += us
        row.append(f"{name} {us:6.1f} us ({2.0 * M * N * K / us / 1e6:5.0f} TF)")
    print(f"{label:16s} N={N:5d} K={K:5d}   " + "   ".join(row))
print("sum: " + "   ".join(f"{k} {v:.1f} us" for k, v in tot.items()))
lib.dvlp_dev_gemm_wide_mode(0)
lib.dvlp_dev_gemm_p8_mode(1)
