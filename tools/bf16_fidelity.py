#!/usr/bin/env python
"""bf16 training fidelity at the benchmark size (B = 64, F = 8, R = 36): N optimisation steps of the bf16 throughput path (hipGraph
replay, fp32 master weights + bf16 shadows) against the fp32 HIP parity path on the SAME batches and start weights, dropout 0.
Prints the per-step losses of both, their deviation, and how far the final parameters drifted apart relative to how far they moved.
    python tools/bf16_fidelity.py [--steps 20] [--lr 1e-5 2e-4] [--batch 64]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demovlp_amd import synthetic as syn  # noqa: E402
from demovlp_amd.loss import GlobalLocalLoss  # noqa: E402
from demovlp_amd.model import ObjectRelation  # noqa: E402
from demovlp_amd.trainer import FusedAdamW, GraphedTrainStep, ParamArena, train_step  # noqa: E402

DEV = "cuda"


def build(F, R, dtype):
    m = ObjectRelation({"model": "", "input_objects": False, "object_num": R, "num_frames": F, "time_module": None},
                       {"model": "pretrained/distilbert-base-uncased", "pretrained": True, "input": "text", "two_outputs": True}, pretrained_init=False, compute_dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(F, R).items()}, strict=True)
    m.set_text_dropout(0.0)
    return m.to(DEV)


def batch(F, R, B, s):
    obj, mask = syn.fast_region_batch(B, F, R, seed=1000 + s)
    ids, att = syn.caption_batch(B, first_sample=s * B)
    return {"text": {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(att).to(DEV)},
            "object": torch.from_numpy(obj).to(DEV), "object_mask": torch.from_numpy(mask).to(DEV)}


def run(dtype, lr, steps, F, R, B, graph, sr=0):
    model = build(F, R, dtype)
    arena = ParamArena(model, bf16_shadow=(dtype == "bfloat16"))
    opt = FusedAdamW(arena, lr=lr)
    if sr and hasattr(opt, "stochastic_shadow"):
        opt.stochastic_shadow = bool(sr)
    p0 = arena.flat_p.clone()
    lf = GlobalLocalLoss(use_local=True, use_global=True, coef=1.0, focal_type="equal")
    stepper = GraphedTrainStep(model, lf, opt, warmup=2) if graph else None
    losses = []
    for s in range(steps):
        out = stepper(batch(F, R, B, s)) if graph else train_step(model, lf, opt, batch(F, R, B, s))
        losses.append([float(x.item()) for x in out])
    torch.cuda.synchronize()
    return np.array(losses), p0, arena.flat_p.clone()


def fidelity(lr, steps=20, F=8, R=36, B=64, sr=0, verbose=True):
    l32, p0, p32 = run("float32", lr, steps, F, R, B, graph=False)
    l16, _, p16 = run("bfloat16", lr, steps, F, R, B, graph=True, sr=sr)
    moved = (p32 - p0).double().norm().item()
    drift = (p16 - p32).double().norm().item()
    cos = torch.nn.functional.cosine_similarity((p16 - p0).double(), (p32 - p0).double(), dim=0).item()
    dev = np.abs(l16[:, 0] - l32[:, 0])
    if verbose:
        print(f"lr {lr:g}, {steps} steps, B = {B}{', stochastic shadow rounding' if sr else ''}")
        for s in range(steps):
            print(f"   step {s + 1:2d}  fp32 {l32[s, 0]:10.5f} (global {l32[s, 1]:9.5f} local {l32[s, 2]:9.5f})   bf16 {l16[s, 0]:10.5f}   |diff| {dev[s]:.2e}  rel {dev[s] / abs(l32[s, 0]):.2e}")
        print(f"   max |loss diff| {dev.max():.3e} (rel {np.max(dev / np.abs(l32[:, 0])):.3e});  loss moved {l32[0, 0] - l32[-1, 0]:+.4f} (fp32) {l16[0, 0] - l16[-1, 0]:+.4f} (bf16)")
        print(f"   parameters: |p32 - p0| {moved:.4e}   |p16 - p32| {drift:.4e} = {drift / moved:.3f} of the distance moved;  cos(update16, update32) {cos:.4f}")
    return dict(l32=l32, l16=l16, max_dev=float(dev.max()), max_rel=float(np.max(dev / np.abs(l32[:, 0]))), drift=drift / moved, cos=cos)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--lr", type=float, nargs="+", default=[1e-5, 2e-4])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--sr", type=int, default=0)
    a = ap.parse_args()
    for lr in a.lr:
        fidelity(lr, a.steps, B=a.batch, sr=a.sr)
