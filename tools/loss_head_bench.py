#!/usr/bin/env python
"""The fused loss-head launch (sim_matrix + NormSoftmax + RWA tail, forward and gradients) at B = 64, bf16: matrix-core form vs the
one-wave-per-entry form (MI355X).   python tools/loss_head_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for B in (64, 32):
    a = torch.randn(B, 256, device=dev, generator=g).bfloat16()
    b = torch.randn(B, 256, device=dev, generator=g).bfloat16()
    xs = torch.rand(B, B, device=dev, generator=g)
    for on in (1, 0):
        ops.call("dvlp_dev_loss_mfma", on)
        ts = []
        for _ in range(20):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.global_local_loss(a, b, xs, 0.05, 20.0, 1, 1, 7)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"B = {B}  matrix cores {'on ' if on else 'off'}: {sorted(ts)[len(ts) // 2]:7.1f} us (incl. ~20 us of output allocations and the launch)")
ops.call("dvlp_dev_loss_mfma", 1)
