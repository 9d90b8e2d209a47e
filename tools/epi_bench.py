#!/usr/bin/env python
"""Where a product's time goes (MI355X): the fc1 shape with each epilogue, and with the timing-only ablations of dvlp_dev_gemm_ablate
(8 = no stores, 16 = no epilogue at all).  python tools/epi_bench.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demovlp_amd import _lib, ops  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

lib = _lib.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ws = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
lib.dvlp_set_workspace(ctypes.c_void_p(ws.data_ptr()), ws.numel())
M = 18496
for N, K in ((3072, 768), (768, 768)):
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    B = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    res = torch.randn(M, N, device=dev, generator=g).bfloat16()
    aux = torch.randn(M, N, device=dev, generator=g).bfloat16()
    for label, b_, r_, a_, flags in (("plain", None, None, None, 0), ("bias", bias, None, None, 0), ("bias+res", bias, res, None, 0),
                                     ("bias+gelu(+aux out)", bias, None, aux, 1), ("gelu_bwd(aux in)", None, None, aux, 2)):
        row = []
        for abl in (0, 8, 16, 32, 64):
            lib.dvlp_dev_gemm_ablate(abl)

            def run():
                rc = lib.dvlp_gemm(1, 0, 0, M, N, K, P(A), K, P(B), K, P(C), N, P(b_), P(r_), N if r_ is not None else 0, P(a_),
                                   N if a_ is not None else 0, flags, 1.0, st)
                assert rc == 0, rc
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                run()
            b.record()
            torch.cuda.synchronize()
            row.append(a.elapsed_time(b) * 50)
        lib.dvlp_dev_gemm_ablate(0)
        fl = 2.0 * M * N * K
        print(f"N={N:5d} K={K:5d} {label:22s} full {row[0]:7.1f} us ({fl / row[0] / 1e6:6.0f} TF)   no stores {row[1]:7.1f}   K loop only {row[2]:7.1f}   stores to 256 rows {row[3]:7.1f}   plain stores {row[4]:7.1f}")
