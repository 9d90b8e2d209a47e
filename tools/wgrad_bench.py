#!/usr/bin/env python
"""The weight-gradient group of one transformer layer (dvlp_wgrad_grouped: fc2, fc1, proj, qkv dW; K = batch x tokens) alone.
--alias: every token row of both operands aliases ONE row (ld = 0), so the whole operand stream is L2-resident: the K loop's
speed with the fabric taken out of the picture (what a perfectly L2-shared schedule could reach)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=18496)
ap.add_argument("--alias", action="store_true")
ap.add_argument("--separate", action="store_true", help="four dvlp_gemm launches instead of the grouped one")
ap.add_argument("--patches", type=int, default=1, help="1 (default): 3 x 3 tile patches pinned to XCDs; 0: per-problem tile order")
ap.add_argument("--subset", default="", help="comma list of problem indices (0 fc2, 1 fc1, 2 proj, 3 qkv)")
a = ap.parse_args()
ops.call("dvlp_dev_wgrad_group_patches", a.patches)
dev = "cuda"
T = a.tokens
g = torch.Generator(device=dev).manual_seed(0)
ops.ensure_gemm_workspace(torch.device(dev), 512)


def mat(rows, cols):
    if a.alias:
        base = torch.randn(1, cols, device=dev, generator=g).bfloat16()
        return torch.as_strided(base, (rows, cols), (0, 1))
    return torch.randn(rows, cols, device=dev, generator=g).bfloat16()


shapes = [("fc2", 768, 3072), ("fc1", 3072, 768), ("proj", 768, 768), ("qkv", 2304, 768)]       # dW [N_out, K_in] = dy[T, N_out]^T x[T, K_in]
probs = [(mat(T, n), mat(T, k), torch.empty(n, k, device=dev, dtype=torch.float32)) for _, n, k in shapes]
if a.subset:
    keep = [int(x) for x in a.subset.split(',')]
    shapes = [shapes[i] for i in keep]
    probs = [probs[i] for i in keep]
flops = sum(2.0 * T * n * k for _, n, k in shapes)


def run():
    if a.separate:
        for dy, x, out in probs:
            ops.linear_bwd_weight(dy, x, out=out)
    else:
        ops.wgrad_grouped(probs)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / n
print(f"wgrad group T={T} alias={a.alias} separate={a.separate} subset={a.subset}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s")
if not a.alias and not a.subset:
    ref = probs[2][0].float().t() @ probs[2][1].float()
    print("   proj dW rel err vs torch fp32:", float((probs[2][2] - ref).abs().max() / ref.abs().max()))
