"""BUTD question-answering head: drop-in mirror of model/video_qa_mdoel.py:78-97 (`BUTDQAHead` and its FCNet / Attention /
SimpleClassifier parts; same sub-module names, so the `head.*` state_dict keys -- weight-normalised linears with `weight_g` /
`weight_v` -- interchange with reference checkpoints).

This head is a downstream classifier on top of the two towers, ~0.5 MFLOP per sample on [B, 256]
vectors: it runs on PyTorch-ROCm's stock kernels; the towers underneath are the hand-written HIP path.  Dropout layers of the
reference head (p = 0.2 in Attention, 0 elsewhere) follow module.train()/eval() as usual.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn.utils import weight_norm


def _fc(dims, norm=True):
    layers = []
    for a, b in zip(dims[:-1], dims[1:]):
        lin = nn.Linear(a, b)
        layers += [weight_norm(lin, dim=None) if norm else lin, nn.ReLU()]
    return layers


class FCNet(nn.Module):
    def __init__(self, dims, dropout=0.0, norm=True):
        super().__init__()
        self.main = nn.Sequential(*_fc(dims, norm))

    def forward(self, x):
        return self.main(x)


class SimpleClassifier(nn.Module):
    def __init__(self, in_dim, hid_dim, out_dim, dropout=0.0):
        super().__init__()
        self.q_net = FCNet([in_dim[0], hid_dim[0]], dropout)
        self.v_net = FCNet([in_dim[1], hid_dim[0]], dropout)
        self.main = nn.Sequential(nn.Linear(hid_dim[0], hid_dim[1]), nn.ReLU(), nn.Dropout(dropout, inplace=True), nn.Linear(hid_dim[1], out_dim))

    def forward(self, q_emb, v_emb):
        return self.main(self.q_net(q_emb) * self.v_net(v_emb))


class Attention(nn.Module):
    def __init__(self, v_dim, q_dim, hid_dim, glimpses=1, dropout=0.2):
        super().__init__()
        self.v_proj = FCNet([v_dim, hid_dim], dropout)
        self.q_proj = FCNet([q_dim, hid_dim], dropout)
        self.drop = nn.Dropout(dropout)
        self.linear = weight_norm(nn.Linear(hid_dim, glimpses), dim=None)

    def forward(self, v, v_mask, q):
        """v [B, k, v_dim], v_mask [B, k] (1 = real region), q [B, q_dim] -> (softmax over k of the masked logits, logits)."""
        logits = self.linear(self.drop(self.v_proj(v) * self.q_proj(q).unsqueeze(1)))
        logits = logits * v_mask.unsqueeze(-1)          # the reference MULTIPLIES by the 0/1 mask (padded regions keep logit 0)
        return torch.softmax(logits, 1), logits


class BUTDQAHead(nn.Module):
    def __init__(self, v_dim, q_dim, hid_dim, out_dim):
        super().__init__()
        self.v_att = Attention(v_dim, q_dim, hid_dim)
        self.classifier = SimpleClassifier([q_dim, v_dim], [hid_dim, hid_dim * 2], out_dim)

    def forward(self, txt_embed, obj_embed, obj_mask):
        att, _ = self.v_att(obj_embed, obj_mask, txt_embed)
        return self.classifier(txt_embed, (att * obj_embed).sum(1))
