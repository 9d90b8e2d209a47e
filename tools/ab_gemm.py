#!/usr/bin/env python
"""A/B of two builds of libdemovlp_hip.so IN ONE PROCESS (boxes differ by 20 % and more, so timings from two gpurun calls cannot be
compared): python tools/ab_gemm.py demovlp_amd/lib/ab_prev.so   -- interleaved rounds, median per shape."""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demovlp_amd import _lib  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

libs = {"new": _lib.load()}
prev = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for name, (ret, argtypes) in _lib._SIGS.items():
    if hasattr(prev, name):
        fn = getattr(prev, name)
        fn.argtypes, fn.restype = argtypes, ret
libs["prev"] = prev
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
ws = {k: torch.empty(256 << 20, device=dev, dtype=torch.uint8) for k in libs}
for k, lib in libs.items():
    lib.dvlp_set_workspace(ctypes.c_void_p(ws[k].data_ptr()), ws[k].numel())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
T = 18496
CASES = [("qkv fwd", 0, 0, T, 2304, 768, "bias"), ("proj fwd", 0, 0, T, 768, 768, "res"), ("fc1 fwd", 0, 0, T, 3072, 768, "gelu"),
         ("fc2 fwd", 0, 0, T, 768, 3072, "res"), ("qkv dX", 0, 1, T, 768, 2304, ""), ("proj dX", 0, 1, T, 768, 768, ""),
         ("fc1 dX", 0, 1, T, 768, 3072, ""), ("fc2 dX(gelu')", 0, 1, T, 3072, 768, "gelu_bwd")]
for label, ta, tb, M, N, K, epi in CASES:
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    B = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16() if not tb else (torch.randn(K, N, device=dev, generator=g) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(N, device=dev) if epi in ("bias", "res", "gelu") else None
    res = torch.randn(M, N, device=dev, generator=g).bfloat16() if epi == "res" else None
    aux = torch.randn(M, N, device=dev, generator=g).bfloat16() if epi in ("gelu", "gelu_bwd") else None
    flags = 1 if epi == "gelu" else 2 if epi == "gelu_bwd" else 0
    ldb = K if not tb else N

    def run(lib):
        rc = lib.dvlp_gemm(1, ta, tb, M, N, K, P(A), K, P(B), ldb, P(C), N, P(bias), P(res), N if res is not None else 0, P(aux),
                           N if aux is not None else 0, flags, 1.0, st)
        assert rc == 0, rc
    times = {k: [] for k in libs}
    outs = {}
    for k, lib in libs.items():
        run(lib)
        torch.cuda.synchronize()
        outs[k] = C.clone()
    for rnd in range(7):
        for k, lib in libs.items():
            for _ in range(2):
                run(lib)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                run(lib)
            b.record()
            torch.cuda.synchronize()
            times[k].append(a.elapsed_time(b) * 100)
    m = {k: statistics.median(v) for k, v in times.items()}
    same = torch.equal(outs["new"], outs["prev"])
    print(f"{label:14s} M={M} N={N} K={K}: prev {m['prev']:7.1f} us  new {m['new']:7.1f} us  ({100 * (m['prev'] / m['new'] - 1):+5.1f} %)  bit-equal={same} max|d|={(outs['new'].float() - outs['prev'].float()).abs().max().item():.3g}")
