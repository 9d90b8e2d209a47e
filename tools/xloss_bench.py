#!/usr/bin/env python
"""Local-loss (K11) forward and backward on the bench shape, generic vs bf16-specialised per-pair kernels (MI355X).
python tools/xloss_bench.py [B G W]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

B, G, W = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 288, 99)
ONLY = int(os.environ["XLOSS_ONLY"]) if "XLOSS_ONLY" in os.environ else None      # profile one variant
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
C = torch.randn(B, G, 256, device=dev, generator=g).bfloat16()
Q = torch.randn(B, W, 256, device=dev, generator=g).bfloat16()
mi = torch.zeros(B, G, device=dev)
mc = torch.zeros(B, W, device=dev)
mc[:, 30:] = -100.0
dsc = torch.randn(B, B, device=dev, generator=g)


def med(fn, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


res = {}
for variant in ((ONLY,) if ONLY is not None else (0, 1, 2)):                 # 0 generic kernels, 1 bf16 kernels with the weighted contexts materialised, 2 bf16 kernels + Gram form
    ops.call("dvlp_dev_xattn_bwd_variant", int(variant > 0))
    ops.call("dvlp_dev_xattn_gram", int(variant == 2))
    sc, ws = ops.xattn_fwd(C, Q, mi, mc, 20.0, True, True)
    keep = ws.clone()
    tf = med(lambda: ops.xattn_fwd(C, Q, mi, mc, 20.0, True, True))

    def bwd():
        ws.copy_(keep)
        return ops.xattn_bwd(C, Q, mi, mc, 20.0, True, dsc, ws)
    tcopy = med(lambda: ws.copy_(keep))
    tb = med(bwd) - tcopy
    dC, dQ = bwd()
    res[variant] = (sc.clone(), dC.float().clone(), dQ.float().clone())
    print(f"variant {variant}: forward {tf:8.1f} us   backward {tb:8.1f} us")
ops.call("dvlp_dev_xattn_bwd_variant", 1)
ops.call("dvlp_dev_xattn_gram", 1)
for k, name in enumerate(("scores", "dC", "dQ") if ONLY is None else ()):
    d = (res[1][k] - res[0][k]).abs().max().item()
    d2 = (res[2][k] - res[0][k]).abs().max().item()
    print(f"{name}: max |bf16 kernels - generic| = {d:.3g}, max |Gram form - generic| = {d2:.3g} (max |generic| = {res[0][k].abs().max().item():.3g})")
