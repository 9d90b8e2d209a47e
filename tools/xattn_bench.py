#!/usr/bin/env python
"""Micro-benchmark of the local-loss pipeline (dvlp_xattn_fwd / bwd) at the bench shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops  # noqa: E402

B, G, W = 64, 288, 99
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
C = torch.randn(B, G, 256, device=dev, generator=g).bfloat16()
Q = torch.randn(B, W, 256, device=dev, generator=g).bfloat16()
mi = torch.zeros(B, G, device=dev)
mc = torch.full((B, W), -100.0, device=dev)
mc[:, :20] = 0
ds = torch.randn(B, B, device=dev, generator=g)
ops.ensure_gemm_workspace(C.device)


def step():
    s, ws = ops.xattn_fwd(C, Q, mi, mc, 20.0, True, True)
    return ops.xattn_bwd(C, Q, mi, mc, 20.0, True, ds, ws)


for _ in range(2):
    step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    step()
b.record()
torch.cuda.synchronize()
print("xattn fwd+bwd: %.3f ms" % (a.elapsed_time(b) / 5))
