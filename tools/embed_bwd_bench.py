#!/usr/bin/env python
"""The word-embedding gradient (dvlp_text_embed_bwd) on the bench's captions: time per launch (MI355X).   python tools/embed_bwd_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops, synthetic as syn  # noqa: E402
from demovlp_amd.ops import call, dt, p, stream  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ids, _ = syn.caption_batch(B, first_sample=0)
ids = torch.from_numpy(ids).to("cuda").reshape(-1)
de = torch.randn(ids.numel(), 768, device="cuda").bfloat16()
dword = torch.zeros((30522, 768), device="cuda", dtype=torch.float32)
ref = torch.zeros_like(dword).index_add_(0, ids, de.float() * (ids != 0).float()[:, None])
call("dvlp_text_embed_bwd", dt(de), de.shape[0], p(ids), p(de), p(dword), stream())
print("max |err| vs index_add:", float((dword - ref).abs().max()), " tokens", ids.numel(), "non-pad", int((ids != 0).sum()), "distinct", int(torch.unique(ids).numel()))
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call("dvlp_text_embed_bwd", dt(de), de.shape[0], p(ids), p(de), p(dword), stream())
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("median %.1f us" % sorted(ts)[len(ts) // 2])
