#!/usr/bin/env python
"""K1 region select alone ("timed separately from raw [F, Nraw, .] buffers"): achieved GB/s against the
algorithmic bytes  read F keep (2048 + 4 + 1) 4 + write F R 2054 4 (+ mask / order)  per sample."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops, synthetic as syn  # noqa: E402

dev = "cuda"
B, F, R = int(os.environ.get("SELECT_B", "64")), int(os.environ.get("SELECT_F", "8")), 36      # SELECT_F=32 SELECT_B=16: BASELINE config 5
for Nraw in (36, 50, 100):
    g = torch.Generator(device=dev).manual_seed(Nraw)
    feats = torch.rand(B, F, Nraw, 2048, device=dev, generator=g)
    bbox = torch.rand(B, F, Nraw, 4, device=dev, generator=g) * 300
    conf = torch.rand(B, F, Nraw, device=dev, generator=g)
    wh = torch.tensor([640.0, 360.0], device=dev).repeat(B, F, 1)
    for _ in range(3):
        ops.region_select(feats, bbox, conf, wh, R)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    a.record()
    for _ in range(n):
        ops.region_select(feats, bbox, conf, wh, R)
    b.record()
    torch.cuda.synchronize()
    us = 1e3 * a.elapsed_time(b) / n
    keep = min(Nraw, R)
    byts = B * F * (keep * (2048 + 4) * 4 + Nraw * 4 + R * 2054 * 4 + R * 8)
    print(f"region_select B={B} F={F} Nraw={Nraw} R={R}: {us:8.1f} us  {byts / 1e6:7.1f} MB algorithmic  {byts / us / 1e3:7.1f} GB/s  ({byts / us / 1e3 / 8000:.2f} of 8 TB/s)")
