#!/usr/bin/env python
"""LayerNorm forward on the object tower's token matrix: generic row-per-wave kernel vs the bf16 D = 768 half-wave kernel (MI355X).
python tools/ln_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for M in (18496, 6400):
    x = torch.randn(M, 768, device=dev, generator=g).bfloat16()
    w = torch.randn(768, device=dev, generator=g)
    b = torch.randn(768, device=dev, generator=g)
    outs = {}
    for mode in (0, 1, 0, 1):
        ops.call("dvlp_dev_layernorm_wide", mode)
        for _ in range(3):
            y = ops.layernorm_fwd(x, w, b, 1e-6)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = ops.layernorm_fwd(x, w, b, 1e-6)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 20
        outs[mode] = y
        print(f"M={M} wide={mode}: {us:6.1f} us  {2 * M * 768 * 2 / us / 1e6:5.2f} TB/s")
    print("  bit-equal y:", torch.equal(outs[0][0], outs[1][0]), " mean/rstd max diff:", (outs[0][2] - outs[1][2]).abs().max().item(), (outs[0][3] - outs[1][3]).abs().max().item())
ops.call("dvlp_dev_layernorm_wide", 1)
