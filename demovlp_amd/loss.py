"""Loss heads: drop-in mirror of model/loss.py (GlobalLocalLoss :10-45, RWALoss :48-116, NormSoftmaxLoss :119-138).

Same constructor signatures and forward contracts; the arithmetic runs in the gfx950 kernels.  ``focal_type``:
'equal' applies the focal gate of loss.py:274-283, anything else ('prob', the constructor default) uses H = 1 exactly
as func_attention_fast does (:251-254).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import functional as Fn
from ._lib import DemoVLPHipError


class NormSoftmaxLoss(nn.Module):
    def __init__(self, temperature=0.05):
        super().__init__()
        self.temperature = temperature

    def forward(self, x):
        return Fn.NormSoftmaxFn.apply(x, self.temperature)


class RWALoss(nn.Module):
    def __init__(self, lambda_softmax=20, focal_type="prob", margin=0, max_violation=False):
        super().__init__()
        self.lambda_softmax = lambda_softmax
        self.focal_type = focal_type
        self.margin = margin
        self.max_violation = max_violation

    def get_sim(self, im, s, im_m, s_l, s_m):
        """[n_img, n_cap] local similarity.  ``s_l`` (caption lengths) is accepted and ignored, as in the reference."""
        if not im.is_cuda:
            raise DemoVLPHipError("RWALoss runs on an MI355X device only (no CPU fallback)")
        if s_m is None:
            s_m = torch.zeros(s.shape[:2], device=s.device)   # reference builds ones (:307-308); constant => same softmax
        if im.dtype != s.dtype:
            s = s.to(im.dtype)
        return Fn.XattnFn.apply(im, s, im_m, s_m, float(self.lambda_softmax), self.focal_type == "equal")

    def get_sim_by_segment(self, img_feats, lang_feats, img_mask, lang_length, cap_mask, segment=8, device="cpu", precision=None):
        """Eval-time full grid (model/loss.py:73-103) -> numpy float64 [n_img, n_txt].  The reference walks 8 x 8 tiles in a Python loop;
        here the grid is processed in row blocks sized by workspace (``segment`` is accepted for signature compatibility).

        ``precision`` (not in the reference's signature): ``None`` follows the embeddings -- fp32 embeddings take the fp32 multi-kernel
        path (the 1e-4 parity path), bf16 embeddings the fused per-pair MFMA kernel (``xfused.hip``: S, both softmaxes, both context
        products and the cosines of a pair stay on chip; G <= 288, W <= 112, else the bf16 multi-kernel path); ``'float32'`` /
        ``'bfloat16'`` force one."""
        if not torch.cuda.is_available():
            raise DemoVLPHipError("get_sim_by_segment needs an MI355X device (no CPU fallback)")
        dev = torch.device("cuda")
        if precision is None:
            precision = "bfloat16" if img_feats.dtype == torch.bfloat16 and lang_feats.dtype == torch.bfloat16 else "float32"
        if precision not in ("float32", "bfloat16"):
            raise ValueError(f"precision must be None, 'float32' or 'bfloat16', got {precision!r}")
        cd = torch.bfloat16 if precision == "bfloat16" else torch.float32
        n_img, n_txt = img_feats.shape[0], lang_feats.shape[0]
        sim = np.zeros((n_img, n_txt))
        la = lang_feats.to(dev).to(cd).contiguous()
        lam = cap_mask.to(dev).float().contiguous()
        G, W = img_feats.shape[1], lang_feats.shape[1]
        per_img = n_txt * (G * (W + 8) * 3 + (W + 8) * 256 + G * 256) * 4          # rough bytes of workspace per image row (multi-kernel path)
        rows = max(1, min(n_img, int(4e9 // max(per_img, 1))))
        with torch.no_grad():
            for i0 in range(0, n_img, rows):
                im = img_feats[i0:i0 + rows].to(dev).to(cd).contiguous()
                imm = img_mask[i0:i0 + rows].to(dev).float().contiguous()
                out = Fn.XattnFn.apply(im, la, imm, lam, float(self.lambda_softmax), self.focal_type == "equal")
                sim[i0:i0 + rows] = out.cpu().numpy()
        return sim

    def forward(self, im, s, im_m, s_l, s_m):
        scores = self.get_sim(im, s, im_m, s_l, s_m)
        return Fn.RWATailFn.apply(scores, float(self.lambda_softmax))


class CrossEntropy(nn.Module):
    """model/loss.py:180-187: the QA fine-tuning loss (configs/ft/*_qa-select.json), on the [B, num_label] logits of ObjectQARelation."""

    def __init__(self):
        super().__init__()
        self.loss = nn.CrossEntropyLoss()

    def forward(self, output, target):
        return self.loss(output, target)


class GlobalLocalLoss(nn.Module):
    def __init__(self, temperature=0.05, lambda_softmax=20, focal_type="prob", margin=0, max_violation=False, use_local=True,
                 use_global=True, coef=1000.0):
        super().__init__()
        self.global_loss = NormSoftmaxLoss(temperature)
        self.local_loss = RWALoss(lambda_softmax, focal_type, margin, max_violation)
        self.use_local = use_local
        self.use_global = use_global
        self.cof_local = coef          # read and never used by the reference (:27)

    def forward(self, global_sim, local_im, local_s, local_im_m, local_s_l, local_s_m):
        if not self.use_local:
            loss = self.global_loss(global_sim)
            return loss, loss, torch.tensor([0.0])
        if not self.use_global:
            loss = self.local_loss(local_im, local_s, local_im_m, local_s_l, local_s_m)
            return loss, torch.tensor([0.0]), loss
        g = self.global_loss(global_sim)
        l = self.local_loss(local_im, local_s, local_im_m, local_s_l, local_s_m)
        return g + l, g, l
