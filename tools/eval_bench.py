#!/usr/bin/env python
"""Retrieval evaluation at MSRVTT test-set size (BASELINE.json configs[3] shape: F=8, R=30, 1000 video-caption pairs, batch 32)
through trainer.evaluate(): tower forwards, global sim_matrix [1000 x 1000], the local grid over all 10^6 pairs (bf16 model: the fused
per-pair kernel; --dtype float32: the fp32 multi-kernel parity path), the reference's addend orientation, R@1/5/10/50 / MedR / MeanR both ways.  Weights are the closed-form synthetic
fill, so the metrics are chance-level; what this measures is the wall time of the path (the reference walks the grid in 8 x 8
tiles from a Python loop, model/loss.py:73-103: 15 625 calls)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demovlp_amd import synthetic as syn  # noqa: E402
from demovlp_amd.loss import GlobalLocalLoss  # noqa: E402
from demovlp_amd.model import ObjectRelation  # noqa: E402
from demovlp_amd.trainer import evaluate  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=1000)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--dtype", default="bfloat16")
a = ap.parse_args()
F, R, dev = 8, 30, "cuda"
model = ObjectRelation({"model": "", "input_objects": False, "object_num": R, "num_frames": F, "time_module": None},
                       {"model": "", "pretrained": True, "input": "text", "two_outputs": True}, pretrained_init=False, compute_dtype=a.dtype)
model.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(F, R).items()})
model.to(dev)
loss_fn = GlobalLocalLoss(use_local=True, use_global=True, coef=1.0, focal_type="equal")


def batches():
    for b0 in range(0, a.pairs, a.batch):
        n = min(a.batch, a.pairs - b0)
        obj, mask = syn.fast_region_batch(n, F, R, seed=100 + b0)
        ids, att = syn.caption_batch(n, first_sample=b0)
        yield {"text": {"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(att).to(dev)},
               "object": torch.from_numpy(obj).to(dev), "object_mask": torch.from_numpy(mask).to(dev)}


evaluate(model, loss_fn, list(batches())[:2])          # warm-up
torch.cuda.synchronize()
bl = list(batches())
t0 = time.perf_counter()
res = evaluate(model, loss_fn, bl)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
m = res["nested_val_metrics"]
print(f"evaluate: {a.pairs} pairs ({a.pairs ** 2} local-similarity pairs), {a.dtype}: {dt:.3f} s  val_loss {res['val_loss']:.4f}")
for k in ("t2v_metrics", "v2t_metrics"):
    print("  ", k, {n: round(float(v), 2) for n, v in m[k].items()})
