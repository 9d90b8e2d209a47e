#!/usr/bin/env python
"""(tile height, K split) sweep of the 256-column ping-pong GEMM kernel on the step's forward / dX shapes: the text tower (M = 6400 tokens) and
the object tower (M = 18496).  For every shape: each height 256 / 224 / 192 / 160 (dvlp_dev_gemm_p8_short_tiles(10 + MIH)) x each K split
(dvlp_dev_gemm_force_split) that gives <= 2.2 rounds of 256 CUs, timed interleaved in one process, and what the dispatch's own choice
(mode 1, split automatic) takes.  The cost model in csrc/gemm.hip (p8_plan) is fitted to this table.

    python tools/tile_sweep.py [text|obj] [label substring ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from demovlp_amd import _lib, ops  # noqa: E402
_lib.use_dev_library()     # developer switches (dvlp_dev_*) exist only in libdemovlp_hip_dev.so

lib = _lib.load()
dev = "cuda"
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
g = torch.Generator(device=dev).manual_seed(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ops.ensure_gemm_workspace(torch.device(dev), 512)
# (label, N, K, transB, flags, operands: b bias, r residual, a aux)
LAYER = [("qkv fwd (bias)", 2304, 768, 0, 0, "b"), ("out fwd (bias+res)", 768, 768, 0, 0, "br"), ("fc1 fwd (gelu, aux out)", 3072, 768, 0, 1, "ba"),
         ("fc2 fwd (bias+res)", 768, 3072, 0, 0, "br"), ("fc2 dX (gelu', aux in)", 3072, 768, 1, 2, "a"), ("fc1 dX", 768, 3072, 1, 0, ""),
         ("out dX", 768, 768, 1, 0, ""), ("qkv dX", 768, 2304, 1, 0, "")]
args = sys.argv[1:]
towers = [t for t in ("text", "obj", "c4") if t in args] or ["text", "obj"]
subs = [a for a in args if a not in ("text", "obj", "c4")]
MS = {"text": 6400, "obj": 18496, "c4": 7712}          # c4: BASELINE config 4, B = 32, F = 8, R = 30 -> 32 x 241 tokens


def timed(run, reps=5, inner=10):
    ts = []
    for _ in range(reps):
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]


for tower in towers:
    M = MS[tower]
    for label, N, K, tb, flags, ops_ in LAYER:
        if subs and not any(s_ in label for s_ in subs):
            continue
        A = torch.randn(M, K, device=dev, generator=g).bfloat16()
        B = ((torch.randn(N, K, device=dev, generator=g) if not tb else torch.randn(K, N, device=dev, generator=g)) * 0.02).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g) if "b" in ops_ else None
        res = torch.randn(M, N, device=dev, generator=g).bfloat16() if "r" in ops_ else None
        aux = torch.randn(M, N, device=dev, generator=g).bfloat16() if "a" in ops_ else None

        def run():
            rc = lib.dvlp_gemm(1, 0, tb, M, N, K, P(A), K, P(B), N if tb else K, P(C), N, P(bias), P(res), N if res is not None else 0, P(aux),
                               N if aux is not None else 0, flags, 1.0, st)
            assert rc == 0, rc
        fl = 2.0 * M * N * K
        lib.dvlp_dev_gemm_p8_short_tiles(1)
        lib.dvlp_dev_gemm_force_split(0)
        lib.dvlp_dev_gemm_p8_persistent(1)
        auto = timed(run)
        rows = []
        lib.dvlp_dev_gemm_p8_persistent(0)
        for mih in (4, 3, 2, 1):
            h = 128 + 32 * mih
            ntile = -(-M // h) * (N // 256)
            for S in (1, 2, 3, 4):
                if S > 1 and (K // 64 < 8 * S or ntile * S > 2.2 * 256 or ntile >= 200):
                    continue
                lib.dvlp_dev_gemm_p8_short_tiles(10 + mih)
                lib.dvlp_dev_gemm_force_split(S if S > 1 else 0)
                if S == 1 and K >= 1024 and ntile < 128:
                    lib.dvlp_dev_gemm_force_split(1)
                rows.append((timed(run), h, S, ntile * S))
        lib.dvlp_dev_gemm_force_split(0)
        lib.dvlp_dev_gemm_p8_short_tiles(1)
        lib.dvlp_dev_gemm_p8_persistent(1)
        best = min(rows)
        print(f"{tower:4s} M={M:5d} {label:24s} N={N:4d} K={K:4d}  dispatch {auto:6.1f} us ({fl / auto / 1e6:5.0f} TF)   best {best[0]:6.1f} us = {best[1]} rows x split {best[2]} ({best[3]} blocks, {fl / best[0] / 1e6:5.0f} TF)")
        print("        " + "  ".join(f"{h}x{S}:{t:.1f}" for t, h, S, _ in rows), flush=True)
