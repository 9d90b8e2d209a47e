"""ObjectTransformer: the region-feature video encoder (mirror of model/object_transformer.py:296-452).

Same constructor arguments, parameter names/shapes (state_dict interchange) and forward contract as the reference;
the arithmetic runs in hand-written gfx950 kernels (functional.py -> ops.py -> libdemovlp_hip.so).  ``time_module`` falsy
(every shipped DemoVLP config: space attention only) and ``'timeattn'`` (:227-234, 252-258: divided space-time attention).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import functional as Fn

EMBED, DEPTH, HIDDEN = 768, 12, 3072
VIT_CHECKPOINT = "pretrained/jx_vit_base_p16_224-80ecf9dd.pth"       # relative to the working directory, as object_transformer.py:480


class _Affine(nn.Module):
    """Holds LayerNorm parameters under the reference's names (weight, bias)."""

    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Linear(nn.Module):
    """Parameters of an nn.Linear under the reference's names.  ``init='linear'``: torch's nn.Linear default (what every Linear of the
    reference's tower starts from when num_frames > 1, and object_embedding / pos_embedding / proj always: they are created after
    ``self.apply(self._init_weights)``, object_transformer.py:364-381); ``init='vit'``: ``_init_weights`` (:384-391: trunc_normal
    std 0.02, zero bias) -- the blocks when num_frames == 1."""

    def __init__(self, fin, fout, bias=True, init="vit"):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.zeros(fout)) if bias else None
        self.reset_parameters(init)

    def reset_parameters(self, init="vit"):
        if init == "vit":
            nn.init.trunc_normal_(self.weight, std=0.02)
            if self.bias is not None:
                nn.init.zeros_(self.bias)
        else:
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
            if self.bias is not None:
                bound = 1.0 / math.sqrt(self.weight.shape[1])
                nn.init.uniform_(self.bias, -bound, bound)


class VarAttention(nn.Module):
    def __init__(self, dim, init="vit"):
        super().__init__()
        self.qkv = _Linear(dim, dim * 3, init=init)
        self.proj = _Linear(dim, dim, init=init)


class Mlp(nn.Module):
    def __init__(self, dim, hidden, init="vit"):
        super().__init__()
        self.fc1 = _Linear(dim, hidden, init=init)
        self.fc2 = _Linear(hidden, dim, init=init)


class SpaceTimeBlock(nn.Module):
    """model/object_transformer.py:199-274 with time_module falsy.  norm3 exists (and never gets a gradient), as in
    the reference, so checkpoints interchange."""

    def __init__(self, dim=EMBED, hidden=HIDDEN, time_module=None, init="vit"):
        super().__init__()
        self.time_module = time_module
        self.norm1 = _Affine(dim)
        self.attn = VarAttention(dim, init)
        if time_module == "timeattn":
            self.timeattn = VarAttention(dim, init)    # time_init='rand' in the reference (:318): ordinary initialisation
        self.norm2 = _Affine(dim)
        self.mlp = Mlp(dim, hidden, init)
        self.norm3 = _Affine(dim)

    def forward(self, x, addmask, frames, regions, addmask_t=None, f2b_below=None, f2b_from_above=False):
        if self.time_module == "timeattn":
            t = self.timeattn
            return Fn.TimeSpaceBlockFn.apply(x, addmask, addmask_t, self.norm3.weight, self.norm3.bias, t.qkv.weight, t.qkv.bias,
                                             t.proj.weight, t.proj.bias, self.norm1.weight, self.norm1.bias, self.attn.qkv.weight,
                                             self.attn.qkv.bias, self.attn.proj.weight, self.attn.proj.bias, self.norm2.weight,
                                             self.norm2.bias, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight,
                                             self.mlp.fc2.bias, frames, regions)
        return Fn.VitBlockFn.apply(x, addmask, self.norm1.weight, self.norm1.bias, self.attn.qkv.weight, self.attn.qkv.bias,
                                   self.attn.proj.weight, self.attn.proj.bias, self.norm2.weight, self.norm2.bias,
                                   self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, frames, regions,
                                   f2b_below, f2b_from_above)


class ObjectTransformer(nn.Module):
    def __init__(self, input_dim=2054, region_nums=20, num_frames=4, output_dim=256, time_module=None):
        super().__init__()
        if time_module and time_module != "timeattn":
            raise NotImplementedError(f"time_module={time_module!r}: the reference knows only falsy and 'timeattn'")
        self.time_module = time_module or None
        if input_dim != 2054:
            raise NotImplementedError("region features are 2048-d + 6-d box geometry")
        self.num_frames = num_frames
        self.embed_dim = self.num_features = EMBED
        self.patches_per_frame = region_nums
        self.feat_dim = 2048
        self.cls_token = nn.Parameter(torch.zeros(1, 1, EMBED))
        self.custom_pos_embed = nn.Parameter(torch.zeros(1, region_nums + 1, EMBED))
        self.temporal_embed = nn.Parameter(torch.zeros(1, num_frames, EMBED))
        # initial values as the reference leaves them BEFORE load_clip_pt_weight overwrites the blocks (:364-381): with one frame
        # `self.apply(self._init_weights)` has run over the blocks (trunc_normal 0.02, zero biases); with more frames the blocks keep
        # nn.Linear's default; the three Linears created after that call keep nn.Linear's default in both cases
        blk_init = "vit" if num_frames == 1 else "linear"
        self.blocks = nn.ModuleList([SpaceTimeBlock(time_module=self.time_module, init=blk_init) for _ in range(DEPTH)])
        self.norm = _Affine(EMBED)            # defined and never applied (object_transformer.py:354, 446-452)
        self.object_embedding = _Linear(self.feat_dim, EMBED, init="linear")
        self.pos_embedding = _Linear(input_dim - self.feat_dim, EMBED, init="linear")
        self.proj = _Linear(EMBED, output_dim, bias=False, init="linear")
        nn.init.trunc_normal_(self.custom_pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.compute_dtype = torch.float32
        self.grad_cut = None           # block index -- or several -- at which backward is cut (data-parallel graph step), None = one piece
        self._cut = None

    def forward_features(self, x, x_mask):
        B, F, R, C = x.shape
        if R != self.patches_per_frame:
            raise ValueError(f"expected {self.patches_per_frame} regions per frame, got {R}")
        if F > self.num_frames:
            raise ValueError(f"{F} frames > temporal_embed size {self.num_frames}")
        self._cut = None
        obj = x.contiguous().float()
        mask01 = x_mask.reshape(B, F, R).contiguous().float()
        tok, addmask = Fn.ObjectPrologueFn.apply(obj, mask01, self.object_embedding.weight, self.object_embedding.bias,
                                                 self.pos_embedding.weight, self.pos_embedding.bias, self.temporal_embed, self.cls_token,
                                                 self.custom_pos_embed, self.compute_dtype)
        addmask_t = None
        if self.time_module == "timeattn":
            from . import ops
            addmask_t = ops.token_transpose(addmask.reshape(B, 1 + F * R, 1), B, F, R).reshape(B, 1 + F * R)     # key mask in region-major order
        cuts = () if self.grad_cut is None else ((self.grad_cut,) if isinstance(self.grad_cut, int) else tuple(self.grad_cut))
        for i, blk in enumerate(self.blocks):
            if i in cuts and tok.requires_grad and torch.is_grad_enabled():
                # backward in pieces (trainer.backward_first / backward_next): autograd stops at this leaf, the caller exchanges the
                # gradients that are final by then, and resumes from the leaf's gradient
                leaf = tok.detach().requires_grad_(True)
                self._cut = (self._cut or []) + [(tok, leaf)]
                tok = leaf
            tok = blk(tok, addmask, F, R, addmask_t, **self._bias_grad_links(i))
        return tok, addmask

    def _bias_grad_links(self, i):
        """Blocks are chained output -> input here, so block i's norm1 backward can emit the fc2-bias gradient of block i - 1 (see
        VitBlockFn.forward).  Only with gradient arenas attached (the sums are deferred reductions into the arena slice)."""
        if self.time_module or not Fn.FUSE_LN_COLSUM or not torch.is_grad_enabled():
            return {}

        def linked(j):       # the pair (block j, block j + 1)
            if j < 0 or j + 1 >= len(self.blocks):
                return False
            b, up = self.blocks[j].mlp.fc2.bias, self.blocks[j + 1].norm1
            return all(getattr(t, "_dvlp_grad_view", None) is not None and t.requires_grad for t in (b, up.weight, up.bias))
        return dict(f2b_below=self.blocks[i - 1].mlp.fc2.bias if linked(i - 1) else None, f2b_from_above=linked(i))

    def take_cut(self):
        """The LAST open cut of the last forward -- (tokens entering that block, the leaf that replaced them) -- or None; pops it.  Cuts
        are resumed from the top of the tower down, one per call."""
        if not self._cut:
            self._cut = None
            return None
        cut = self._cut.pop()
        if not self._cut:
            self._cut = None
        return cut

    def forward(self, x, x_mask):
        """x [B,F,R,2054], x_mask [B,F,R] (1 = real region) -> (embeddings [B,N,256], additive mask [B,N])."""
        tok, addmask = self.forward_features(x, x_mask)
        return Fn.LinearFn.apply(tok, self.proj.weight, None, None), addmask


def load_clip_pt_weight(model, path=VIT_CHECKPOINT):
    """model/object_transformer.py:470-483: initialise the tower from timm's ViT-B/16 checkpoint, ``strict=False`` -- every
    ``blocks.N.{norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2}``, ``cls_token`` and ``norm`` tensor matches by name and is loaded;
    ``pos_embed``, ``patch_embed.*`` and ``head.*`` have no counterpart and are ignored; ``custom_pos_embed``, ``temporal_embed``,
    ``norm3``, ``object_embedding``, ``pos_embedding``, ``proj`` (and ``timeattn``) keep their initial values.  A missing file raises
    ``FileNotFoundError`` and a tensor of the wrong shape ``RuntimeError``, as in the reference."""
    vit_checkpoint = torch.load(path, map_location="cpu")
    model.load_state_dict(vit_checkpoint, strict=False)
    return model
