#!/usr/bin/env python
"""The batched contractions of the local loss (B = 64, G = 288, W = 99 -> Wp = 104, d = 256) on this library's GEMM against the stock
library (torch.bmm -> hipBLASLt) and against their HBM bound (operands read once + output written once at 5 TB/s): they are skinny
streaming products, not MFMA-bound ones.      python tools/loss_gemm_bench.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from demovlp_amd import _lib, ops  # noqa: E402
_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

lib = _lib.load()
dev = "cuda"
ops.ensure_gemm_workspace(torch.device(dev), 512)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


# (label, transA, transB, M, N, K, batch): C[M,N] = op(A) op(B)^T with A stored [M,K] (or [K,M] if transA), B stored [N,K] (or [K,N] if transB)
shapes = [("wc = P1 C^        (KR)", 0, 1, 6656, 256, 288, 64), ("dP1 = dwc C^T    (KK)", 0, 0, 6656, 288, 256, 64), ("T = P2 Kq        (KK)", 0, 0, 18432, 104, 104, 64),
          ("dC^ += P1^T dwc  (RR)", 1, 1, 288, 256, 6656, 64), ("dC^ += dS Q^     (KR)", 0, 1, 288, 256, 6656, 64), ("dQ^ = dS^T C^    (RR)", 1, 1, 104, 256, 18432, 64),
          ("dKq = T^T P2     (RR)", 1, 1, 104, 104, 18432, 64), ("S = C^ Q^T       (KK)", 0, 0, 18432, 6656, 256, 1)]
for label, ta, tb, M, N, K, nb in shapes:
    A = torch.randn((nb, K, M) if ta else (nb, M, K), device=dev, generator=g).to(bf)
    B = torch.randn((nb, K, N) if tb else (nb, N, K), device=dev, generator=g).to(bf)
    C = torch.empty(nb, M, N, device=dev, dtype=bf)

    def ours():
        rc = lib.dvlp_gemm_batched(1, ta, tb, M, N, K, P(A), M if ta else K, P(B), N if tb else K, P(C), N, None, None, 0, None, 0, 0, 1.0, nb,
                                   A[0].numel(), B[0].numel(), M * N, 0, 0, st)
        assert rc == 0, rc
    Am = A.transpose(1, 2) if ta else A
    Bm = B if tb else B.transpose(1, 2)

    def stock():
        torch.bmm(Am, Bm, out=C)
    lib.dvlp_dev_gemm_resident_b(0)
    t0 = bench(ours)                  # the tile kernels (round 5's path)
    lib.dvlp_dev_gemm_resident_b(1)
    t1, t2 = bench(ours), bench(stock)
    byts = 2.0 * nb * (M * K + N * K + M * N)
    fl = 2.0 * nb * M * N * K
    print(f"{label}  M={M:6d} N={N:5d} K={K:6d} b={nb:3d}   tile kernel {t0:7.1f} us   ours {t1:7.1f} us ({fl / t1 / 1e6:5.0f} TF, {byts / t1 / 1e6:5.2f} TB/s)   stock {t2:7.1f} us ({fl / t2 / 1e6:5.0f} TF)"
          f"   HBM bound {byts / 5e6:6.1f} us")
