#!/usr/bin/env python
"""The local loss' long-K batched reductions (dC^_i, dQ^_j, dKq_j) and its all-pairs product S on the 256 x 256 ping-pong kernel with forced K
splits / tile heights against the default dispatch -- the table behind the `longk8` rule and the persistent-form rule of csrc/gemm.hip
(profiles/r6_loss_reductions_on_p8.txt).      python tools/loss_reduction_sweep.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from demovlp_amd import _lib, ops
_lib.use_dev_library()
lib = _lib.load()
dev = "cuda"
ops.ensure_gemm_workspace(torch.device(dev), 512)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
shapes = [("dC^ += P1^T dwc  (RR)", 1, 1, 288, 256, 6656, 64, 0), ("dC^ += dS Q^     (KR)", 0, 1, 288, 256, 6656, 64, 8), ("dQ^ = dS^T C^    (RR)", 1, 1, 104, 256, 18432, 64, 0),
          ("dKq = T^T P2     (RR)", 1, 1, 104, 104, 18432, 64, 0), ("S = C^ Q^T (KK)", 0, 0, 18432, 6656, 256, 1, 16)]
for label, ta, tb, M, N, K, nb, flags in shapes:
    A = torch.randn((nb, K, M) if ta else (nb, M, K), device=dev, generator=g).to(bf)
    B = torch.randn((nb, K, N) if tb else (nb, N, K), device=dev, generator=g).to(bf)
    C = torch.zeros(nb, M, N, device=dev, dtype=bf)
    def ours():
        rc = lib.dvlp_gemm_batched(1, ta, tb, M, N, K, P(A), M if ta else K, P(B), N if tb else K, P(C), N, None, None, 0, None, 0, flags & ~8, 1.0, nb,
                                   A[0].numel(), B[0].numel(), M * N, 0, 0, st)
        assert rc == 0, rc
    lib.dvlp_dev_gemm_p8_mode(1); lib.dvlp_dev_gemm_force_split(0); lib.dvlp_dev_gemm_p8_short_tiles(1)
    base = bench(ours); ref = C.clone()
    out = [f"default {base:6.1f}"]
    for mode, sp, sh in ((2, 0, 1), (2, 1, 1), (2, 2, 1), (2, 4, 1), (2, 0, 2)):
        lib.dvlp_dev_gemm_p8_mode(mode); lib.dvlp_dev_gemm_force_split(sp); lib.dvlp_dev_gemm_p8_short_tiles(sh)
        try:
            t = bench(ours)
            err = float((C.float() - ref.float()).abs().max() / ref.float().abs().max())
            out.append(f"p8 split={sp or 'auto'} short={sh}: {t:6.1f} (dev {err:.1e})")
        except AssertionError as e:
            out.append(f"p8 split={sp}: rc {e}")
    print(label, f"M={M} N={N} K={K} b={nb}:", " | ".join(out), flush=True)
lib.dvlp_dev_gemm_p8_mode(1); lib.dvlp_dev_gemm_force_split(0); lib.dvlp_dev_gemm_p8_short_tiles(1)
