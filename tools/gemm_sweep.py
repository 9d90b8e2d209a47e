#!/usr/bin/env python
"""Sweep kernel choice (256 x 256 "p8" vs 128 x 128 "glds") and K split for the hot path's under-filled GEMM shapes: the text tower
(M = 64 x 100 tokens), the two 256-wide projections and the region-embedding products.  Prints us and TFLOP/s per configuration --
the dispatch heuristics in csrc/gemm.hip (dvlp_gemm_batched) are set from tables like this one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
ops.ensure_gemm_workspace(torch.device(dev), 512)


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


def rnd(*shape):
    return torch.randn(*shape, device=dev, generator=g).bfloat16()


# (label, kind, M(tokens), N(out features), K(in features)); kind: fwd = x W^T, dx = dy W, dw = dy^T x
CASES = [("text qkv fwd", "fwd", 6400, 2304, 768), ("text out fwd", "fwd", 6400, 768, 768), ("text fc1 fwd", "fwd", 6400, 3072, 768),
         ("text fc2 fwd", "fwd", 6400, 768, 3072), ("text qkv dX", "dx", 6400, 2304, 768), ("text out dX", "dx", 6400, 768, 768),
         ("text fc1 dX", "dx", 6400, 3072, 768), ("text fc2 dX", "dx", 6400, 768, 3072),
         ("obj proj256 fwd", "fwd", 18496, 256, 768), ("obj proj256 dX", "dx", 18496, 256, 768), ("obj proj256 dW", "dw", 18496, 256, 768),
         ("txt proj256 fwd", "fwd", 6400, 256, 768), ("txt proj256 dX", "dx", 6400, 256, 768), ("txt proj256 dW", "dw", 6400, 256, 768),
         ("embed fwd", "fwd", 18432, 768, 2048), ("embed dW", "dw", 18432, 768, 2048),
         ("obj qkv fwd", "fwd", 18496, 2304, 768), ("obj out fwd", "fwd", 18496, 768, 768), ("obj fc1 fwd", "fwd", 18496, 3072, 768),
         ("obj fc2 fwd", "fwd", 18496, 768, 3072), ("obj qkv dX", "dx", 18496, 2304, 768), ("obj out dX", "dx", 18496, 768, 768),
         ("obj fc1 dX", "dx", 18496, 3072, 768), ("obj fc2 dX", "dx", 18496, 768, 3072)]
only = sys.argv[1:] 
for label, kind, T, N, K in CASES:
    if only and not any(o in label for o in only):
        continue
    x, w, dy = rnd(T, K), rnd(N, K) * 0.02, rnd(T, N)
    pre = rnd(T, N)
    fl = 2.0 * T * N * K
    if kind == "fwd":
        fn = (lambda: ops.linear_fwd(x, w, None, gelu_aux=pre)) if "fc1" in label else (lambda: ops.linear_fwd(x, w))
    elif kind == "dx":
        if "fc2" in label:
            prek = rnd(T, K)
            fn = lambda: ops.linear_bwd_input(dy, w, gelu_pre=prek)       # dx of fc2 carries gelu'(pre of fc1): the epi=2 launch
        else:
            fn = lambda: ops.linear_bwd_input(dy, w)
    else:
        out = torch.empty(N, K, device=dev, dtype=torch.float32)
        fn = lambda: ops.linear_bwd_weight(dy, x, out=out)
    res = []
    for p8 in (0, 2, "w"):                                   # 128 x 128, 256 x 256, 256 x 128 ("wide")
        for S in (0, 1, 2, 3, 4, 6, 8):
            ops.call("dvlp_dev_gemm_p8_mode", 0 if p8 == "w" else p8)
            ops.call("dvlp_dev_gemm_wide_mode", 2 if p8 == "w" else 0)
            ops.call("dvlp_dev_gemm_force_split", S)
            try:
                t = bench(fn)
                res.append((t, p8, S))
            except Exception as e:  # noqa: BLE001
                pass
    ops.call("dvlp_dev_gemm_p8_mode", 1)
    ops.call("dvlp_dev_gemm_wide_mode", 0)
    ops.call("dvlp_dev_gemm_force_split", 0)
    t_auto = bench(fn)
    res.sort()
    best = ", ".join(f"p8={p} S={s}: {t * 1e6:.1f}us" for t, p, s in res[:4])
    print(f"{label:18s} T={T} N={N} K={K}: auto {t_auto * 1e6:6.1f} us ({fl / t_auto / 1e12:6.1f} TF) | best: {best}")
