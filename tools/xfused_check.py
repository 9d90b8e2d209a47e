#!/usr/bin/env python
"""Fused per-pair local-loss kernels against the multi-kernel path and the fp32 oracle, plus timing (developer tool)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import ops  # noqa: E402
from demovlp_amd import _lib as _dvlp_lib  # noqa: E402
_dvlp_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so
from oracle import restatement as orc  # noqa: E402

dev = "cuda"


def case(B, G, W, seed=0, gate=True, bwd=False):
    rng = np.random.default_rng(seed)
    im = rng.standard_normal((B, G, 256), dtype=np.float32)
    cap = rng.standard_normal((B, W, 256), dtype=np.float32)
    n = min(G, W)
    cap[:, :n, :64] += im[:, :n, :64] * 0.5
    m_img = np.zeros((B, G), np.float32)
    if B > 1:
        m_img[1, max(0, G - 5):] = -100.0
    lens = rng.integers(5, min(30, W), B)
    m_cap = np.full((B, W), -100.0, np.float32)
    for b in range(B):
        m_cap[b, : lens[b]] = 0.0
    t = lambda a: torch.from_numpy(a).to(dev)
    C, Q = t(im).bfloat16(), t(cap).bfloat16()
    res = {}
    for mode in (0, 1):
        ops.call("dvlp_dev_xattn_fused_mode", mode)
        s, ws = ops.xattn_fwd(C, Q, t(m_img), t(m_cap), 20.0, gate, bwd)
        res[mode] = s.cpu().numpy()
        if bwd:
            ds = t(rng.standard_normal((B, B), dtype=np.float32))
            dC, dQ = ops.xattn_bwd(C, Q, t(m_img), t(m_cap), 20.0, gate, ds, ws)
            res[(mode, "dC")], res[(mode, "dQ")] = dC.float().cpu().numpy(), dQ.float().cpu().numpy()
    ref = orc.xattn_scores_batched(C.float().cpu(), Q.float().cpu(), torch.from_numpy(m_img), torch.from_numpy(m_cap), 20.0, gate).numpy()
    e0, e1 = np.abs(res[0] - ref).max(), np.abs(res[1] - ref).max()
    print(f"B={B} G={G} W={W} gate={gate}: |multi - oracle| {e0:.2e}  |fused - oracle| {e1:.2e}  |fused - multi| {np.abs(res[0]-res[1]).max():.2e}  (scores ~{np.abs(ref).mean():.3f})")
    if bwd:
        for k in ("dC", "dQ"):
            a, b = res[(0, k)], res[(1, k)]
            print(f"   {k}: rel L2 fused vs multi {np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-12):.3e}")
    return max(e1, 0)


def timing(B=64, G=288, W=99, bwd=False, stops=(0,)):
    g = torch.Generator(device=dev).manual_seed(0)
    C = torch.randn(B, G, 256, device=dev, generator=g).bfloat16()
    Q = torch.randn(B, W, 256, device=dev, generator=g).bfloat16()
    mi = torch.zeros(B, G, device=dev)
    mc = torch.full((B, W), -100.0, device=dev)
    mc[:, :20] = 0
    ds = torch.randn(B, B, device=dev, generator=g)
    ops.ensure_gemm_workspace(C.device)
    for mode, stop in [(0, 0)] + [(1, st) for st in stops]:
        ops.call("dvlp_dev_xattn_fused_mode", mode)
        ops.call("dvlp_dev_xfused_ablate", stop)

        def step():
            s, ws = ops.xattn_fwd(C, Q, mi, mc, 20.0, True, bwd)
            if bwd:
                ops.xattn_bwd(C, Q, mi, mc, 20.0, True, ds, ws)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            step()
        b.record()
        torch.cuda.synchronize()
        print(f"mode {mode} stop {stop} bwd={bwd}: {a.elapsed_time(b) / 5:.3f} ms")
    ops.call("dvlp_dev_xfused_ablate", 0)


if __name__ == "__main__":
    bwd = "--bwd" in sys.argv
    for (B, G, W) in ((2, 288, 99), (4, 288, 99), (3, 240, 99), (3, 30, 99), (2, 16, 99), (5, 48, 37), (2, 288, 112), (3, 100, 7)):
        case(B, G, W, seed=B + G, bwd=bwd)
    case(4, 288, 99, seed=5, gate=False, bwd=bwd)
    timing(bwd=False, stops=(0, 1, 2, 3, 4, 5) if '--ablate' in sys.argv else (0,))
    if bwd:
        timing(bwd=True)
