#!/usr/bin/env python
"""Stage costs of the bf16 per-pair backward kernel of the local loss (MI355X): time of dvlp_xattn_bwd with the kernel cut after each
stage (dvlp_dev_xattn_bwd_stop); differences between rows are the stages.  python tools/xbwd_stages.py [B G W]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

B, G, W = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 288, 99)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
C = torch.randn(B, G, 256, device=dev, generator=g).bfloat16()
Q = torch.randn(B, W, 256, device=dev, generator=g).bfloat16()
mi = torch.zeros(B, G, device=dev)
mc = torch.zeros(B, W, device=dev)
mc[:, 30:] = -100.0
dsc = torch.randn(B, B, device=dev, generator=g)
for variant, stops in ((0, (0,)), (1, (0, 6, 5, 1, 2, 3))):
    ops.call("dvlp_dev_xattn_bwd_variant", variant)
    _, ws = ops.xattn_fwd(C, Q, mi, mc, 20.0, True, True)          # the workspace layout depends on the variant
    keep = ws.clone()
    for stop in stops:
        ops.call("dvlp_dev_xattn_bwd_stop", stop)
        ts = []
        for rep in range(5):
            ws.copy_(keep)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.xattn_bwd(C, Q, mi, mc, 20.0, True, dsc, ws)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        print(f"variant {variant} stop {stop}: whole backward {sorted(ts)[len(ts) // 2]:8.1f} us")
ops.call("dvlp_dev_xattn_bwd_stop", 0)
ops.call("dvlp_dev_xattn_bwd_variant", 1)

# forward kernel (same stops): time of dvlp_xattn_fwd with the softmax kernel cut after each stage
for stop in (0, 6, 5, 1, 2, 7):          # 7: everything but the P1 / P2 stores of the softmax kernel
    ops.call("dvlp_dev_xattn_bwd_stop", stop)
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.xattn_fwd(C, Q, mi, mc, 20.0, True, True)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"forward stop {stop}: whole forward {sorted(ts)[len(ts) // 2]:8.1f} us")
ops.call("dvlp_dev_xattn_bwd_stop", 0)
