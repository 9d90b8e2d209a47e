#!/usr/bin/env python
"""Micro-benchmark of the object tower's space attention (forward, backward) at the bench shape.  Usage: python tools/attn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so


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
    return a.elapsed_time(b) / iters * 1e3


def main():
    if len(sys.argv) > 1:
        ops.call("dvlp_dev_attention_ablate", int(sys.argv[1]))
    B, F, R = 64, 8, 36
    N = 1 + F * R
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * N, 2304, device="cuda", generator=g).to(torch.bfloat16)
    dout = torch.randn(B * N, 768, device="cuda", generator=g).to(torch.bfloat16)
    mask = torch.zeros(B, N, device="cuda")
    mb = 2.0 * B * N * 2304 / 1e6
    persist = [int(x) for x in os.environ.get("ATTN_LEAN", "0,1").split(",")]
    ref = {}
    for fold, per in [(0, persist[0])] + [(1, p) for p in persist] * 2:
        ops.call("dvlp_dev_attention_cls_fold", fold)
        ops.call("dvlp_dev_attention_lean", per)
        out, stats = ops.space_attention_fwd(qkv, mask, B, F, R, want_stats=True)
        dqkv = ops.space_attention_bwd(qkv, mask, dout, B, F, R, out=out, stats=stats)
        torch.cuda.synchronize()
        if fold in ref:     # the round-5 kernels compute the same numbers
            print("   max |out - out(lean %d)| = %.3g   max |dqkv - ...| = %.3g" % (ref[fold][2], (out.float() - ref[fold][0].float()).abs().max().item(),
                                                                                     (dqkv.float() - ref[fold][1].float()).abs().max().item()))
        else:
            ref[fold] = (out.clone(), dqkv.clone(), per)
        t = bench(lambda: ops.space_attention_fwd(qkv, mask, B, F, R, want_stats=True))
        print("CLS fold %d lean %d: space attention fwd  %7.1f us   (qkv read %.0f MB + out %.0f MB -> %.2f TB/s)" % (fold, per, t, mb, mb / 3, (mb + mb / 3) / t))
        t = bench(lambda: ops.space_attention_bwd(qkv, mask, dout, B, F, R, out=out, stats=stats))
        print("CLS fold %d lean %d: space attention bwd  %7.1f us   (qkv + dout read, dqkv written: %.0f MB -> %.2f TB/s)" % (fold, per, t, 2 * mb + mb / 3, (2 * mb + mb / 3) / t))

def traffic_only():
    """The lean forward's loads and stores without its arithmetic (dvlp_dev_attention_ablate bit 8): what the access shape alone costs."""
    B, F, R = 64, 8, 36
    N = 1 + F * R
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * N, 2304, device="cuda", generator=g).to(torch.bfloat16)
    mask = torch.zeros(B, N, device="cuda")
    mb = 2.0 * B * N * 2304 / 1e6
    for abl in (0, 8, 0, 8):
        ops.call("dvlp_dev_attention_ablate", abl)
        t = bench(lambda: ops.space_attention_fwd(qkv, mask, B, F, R, want_stats=True))
        print("ablate %d: lean forward %7.1f us  (%.2f TB/s of its 113 MB)" % (abl, t, (mb + mb / 3) / t))
    ops.call("dvlp_dev_attention_ablate", 0)


if __name__ == "__main__":
    if os.environ.get("ATTN_TRAFFIC_ONLY"):
        traffic_only()
        sys.exit(0)
    main()
