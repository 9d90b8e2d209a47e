"""DistilBERT encoder with HuggingFace's parameter names (``text_model.*`` keys of the reference state_dict), computed
by the gfx950 kernels.  The reference gets this tower from ``AutoModel.from_pretrained`` (model/model.py:29); the
arithmetic is third-party (transformers 4.10.0) -- see SURVEY.md section 8(c).  In train mode (the reference leaves the
text model in train mode, model/model.py:29-30) HuggingFace's three dropouts are applied with the config's ``dropout`` /
``attention_dropout`` (0.1 in distilbert-base-uncased): after the embedding LayerNorm, on the attention probabilities,
after the feed-forward's second linear.  Masks are Philox4x32-10 streams from a device-resident state (csrc/dropout.hip);
``set_dropout(0, 0)`` (what the parity fixtures use) or ``eval()`` turns them off.
"""
from __future__ import annotations

import json
import os

import torch
import torch.nn as nn

from . import functional as Fn
from .object_transformer import _Affine, _Linear


class _Embeddings(nn.Module):
    def __init__(self, vocab, max_pos, dim):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, dim, padding_idx=0)
        self.position_embeddings = nn.Embedding(max_pos, dim)
        self.LayerNorm = _Affine(dim)
        nn.init.normal_(self.word_embeddings.weight, std=0.02)
        nn.init.normal_(self.position_embeddings.weight, std=0.02)


class _Attention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.q_lin, self.k_lin, self.v_lin, self.out_lin = (_Linear(dim, dim) for _ in range(4))


class _FFN(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.lin1 = _Linear(dim, hidden)
        self.lin2 = _Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.attention = _Attention(dim)
        self.sa_layer_norm = _Affine(dim)
        self.ffn = _FFN(dim, hidden)
        self.output_layer_norm = _Affine(dim)

    def forward(self, x, addmask, want_relu=False, drop=None):
        a, f = self.attention, self.ffn
        return Fn.BertLayerFn.apply(x, addmask, a.q_lin.weight, a.q_lin.bias, a.k_lin.weight, a.k_lin.bias, a.v_lin.weight,
                                    a.v_lin.bias, a.out_lin.weight, a.out_lin.bias, self.sa_layer_norm.weight,
                                    self.sa_layer_norm.bias, f.lin1.weight, f.lin1.bias, f.lin2.weight, f.lin2.bias,
                                    self.output_layer_norm.weight, self.output_layer_norm.bias, want_relu, drop)


class _Transformer(nn.Module):
    def __init__(self, n_layers, dim, hidden):
        super().__init__()
        self.layer = nn.ModuleList([_Block(dim, hidden) for _ in range(n_layers)])


class _Config:
    def __init__(self, **kw):
        self.vocab_size = kw.get("vocab_size", 30522)
        self.max_position_embeddings = kw.get("max_position_embeddings", 512)
        self.dim = self.hidden_size = kw.get("dim", 768)
        self.hidden_dim = kw.get("hidden_dim", 3072)
        self.n_layers = kw.get("n_layers", 6)
        self.n_heads = kw.get("n_heads", 12)
        self.dropout = float(kw.get("dropout", 0.1))                        # HF DistilBertConfig defaults
        self.attention_dropout = float(kw.get("attention_dropout", 0.1))


class DistilBertEncoder(nn.Module):
    def __init__(self, config: _Config | None = None):
        super().__init__()
        self.config = config or _Config()
        c = self.config
        if c.dim != 768 or c.n_heads != 12 or c.hidden_dim != 3072:
            raise NotImplementedError("kernels are specialised for distilbert-base (768 / 12 heads / 3072)")
        self.embeddings = _Embeddings(c.vocab_size, c.max_position_embeddings, c.dim)
        self.transformer = _Transformer(c.n_layers, c.dim, c.hidden_dim)
        self.compute_dtype = torch.float32
        self._drop_state = None
        self.dropout_seed = 0

    def set_dropout(self, dropout, attention_dropout=None):
        """Override the config's probabilities (0 / 0: what the parity fixtures were generated with)."""
        self.config.dropout = float(dropout)
        self.config.attention_dropout = float(dropout if attention_dropout is None else attention_dropout)
        return self

    def seed_dropout(self, seed):
        self.dropout_seed, self._drop_state = int(seed), None
        return self

    def dropout_state(self, device):
        from . import ops
        if self._drop_state is None or self._drop_state.device != device:
            self._drop_state = ops.dropout_state(device, self.dropout_seed)
        return self._drop_state

    @classmethod
    def from_pretrained(cls, path, allow_random_init=False):
        """``AutoModel.from_pretrained(text_params['model'])`` (model/model.py:29) for a LOCAL HuggingFace directory: ``config.json``
        + ``model.safetensors`` or ``pytorch_model.bin``; keys with or without the ``distilbert.`` prefix.  As with HuggingFace, a
        missing directory, a missing ``config.json`` or missing weights raise ``OSError`` -- nothing is initialised at random behind the
        caller's back.  ``allow_random_init=True`` (synthetic runs: tests, ``bench.py``, boxes without the checkpoint) builds the
        default distilbert-base-uncased shape with random weights when ``path`` is empty or absent."""
        if not path or not os.path.isdir(path):
            if allow_random_init:
                return cls(_Config())
            raise OSError(f"Can't load config for {path!r}: not a local directory with a config.json (this build has no hub access; "
                          f"pass pretrained_init=False to ObjectRelation for random weights)")
        cj = os.path.join(path, "config.json")
        if not os.path.exists(cj):
            raise OSError(f"Can't load config for {path!r}: {cj} is missing")
        with open(cj) as fh:
            cfg = json.load(fh)
        if cfg.get("model_type", "distilbert") != "distilbert":
            raise NotImplementedError(f"text model type {cfg.get('model_type')!r}: the kernels implement DistilBERT (every shipped config)")
        st = os.path.join(path, "model.safetensors")
        pb = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(pb):
            sd = torch.load(pb, map_location="cpu")
        else:
            raise OSError(f"Error no file named model.safetensors or pytorch_model.bin found in directory {path}.")
        m = cls(_Config(**cfg))
        sd = {k[len("distilbert."):] if k.startswith("distilbert.") else k: v for k, v in sd.items()}
        own = m.state_dict()
        missing = sorted(k for k in own if k not in sd)
        if missing:                       # HuggingFace warns "newly initialized"; a half-loaded tower is never what a caller wants here
            raise OSError(f"checkpoint in {path} lacks {len(missing)} DistilBERT tensors, e.g. {missing[:3]}")
        m.load_state_dict({k: sd[k] for k in own}, strict=True)          # vocab_projector.* / vocab_transform.* (the MLM head) are ignored, as AutoModel does
        return m

    def forward(self, input_ids=None, attention_mask=None, want_relu=False, **_):
        """Returns last_hidden_state [B,L,768] (and relu of it when ``want_relu``)."""
        B, L = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if attention_mask.is_cuda and attention_mask.dtype == torch.int64 and attention_mask.is_contiguous():
            from . import ops
            addmask = ops.text_key_mask(attention_mask)           # one launch (a fill, a comparison and a masked fill otherwise)
        else:
            addmask = torch.zeros((B, L), device=input_ids.device, dtype=torch.float32)
            addmask.masked_fill_(attention_mask == 0, float("-inf"))
        e = self.embeddings
        c = self.config
        dropping = self.training and (c.dropout > 0.0 or c.attention_dropout > 0.0)
        state = None
        if dropping:
            from . import ops
            state = self.dropout_state(input_ids.device)
            ops.dropout_advance(state)                       # one new Philox offset per forward (on the device: graph-replayable)
        x = Fn.TextEmbedFn.apply(input_ids.contiguous(), e.word_embeddings.weight, e.position_embeddings.weight, e.LayerNorm.weight,
                                 e.LayerNorm.bias, self.compute_dtype, (c.dropout, state, 0) if dropping and c.dropout > 0.0 else None)
        n = len(self.transformer.layer)
        xr = None
        for i, blk in enumerate(self.transformer.layer):
            x, xr = blk(x, addmask, want_relu and i == n - 1, (c.attention_dropout, c.dropout, state, 1 + 2 * i) if dropping else None)
        return (x, xr) if want_relu else x
