"""ObjectRelation two-tower model and sim_matrix: drop-in mirror of model/model.py:12-197, 582-590.

Constructor arguments, ``forward(data) -> dict`` contract, attribute names and the 280-tensor ``state_dict`` schema are
the reference's, so ``ConfigParser.initialize('arch', module)`` of a DemoVLP trainer can
build it from the same JSON.  Extra keyword ``compute_dtype`` ('float32' parity path | 'bfloat16' throughput path).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as Fn
from ._lib import DemoVLPHipError
from .object_transformer import ObjectTransformer, load_clip_pt_weight
from .text_model import DistilBertEncoder

_DTYPES = {"float32": torch.float32, "fp32": torch.float32, "bfloat16": torch.bfloat16, "bf16": torch.bfloat16,
           torch.float32: torch.float32, torch.bfloat16: torch.bfloat16}


def state_dict_data_parallel_fix(load_state_dict, curr_state_dict):
    """utils/util.py:30-56: add or strip the 'module.' prefix so checkpoints move between DDP-wrapped and bare models."""
    load_keys, curr_keys = list(load_state_dict.keys()), list(curr_state_dict.keys())
    redo_dp = undo_dp = False
    if not curr_keys[0].startswith("module.") and load_keys[0].startswith("module."):
        undo_dp = True
    elif curr_keys[0].startswith("module.") and not load_keys[0].startswith("module."):
        redo_dp = True
    if undo_dp:
        return {k[7:]: v for k, v in load_state_dict.items()}
    if redo_dp:
        return {"module." + k: v for k, v in load_state_dict.items()}
    return load_state_dict


class ObjectRelation(nn.Module):
    """``pretrained_init=True`` (default) is the reference's constructor (model/model.py:29-36): DistilBERT from the local
    HuggingFace directory ``text_params['model']`` and the object tower from ``pretrained/jx_vit_base_p16_224-80ecf9dd.pth``
    (``load_clip_pt_weight``, strict=False); either one missing raises, as it does there.  ``pretrained_init=False`` is the explicit
    opt-out for synthetic runs (tests, ``bench.py``): both towers start from random weights of the same distributions."""

    def __init__(self, object_params, text_params, projection_dim=256, load_checkpoint=None, projection="minimal",
                 load_temporal_fix="zeros", compute_dtype="float32", pretrained_init=True):
        super().__init__()
        self.text_params = text_params
        self.object_params = object_params
        self.load_temporal_fix = load_temporal_fix
        if not text_params["pretrained"]:
            raise NotImplementedError("Huggingface text models require pretrained init.")
        if projection_dim != 256:
            raise NotImplementedError("projection_dim is 256 on this path (model/model.py:65)")
        self.text_model = DistilBertEncoder.from_pretrained(text_params.get("model"), allow_random_init=not pretrained_init)
        self.text_model.train()                                                                   # model/model.py:30
        self.object_model = ObjectTransformer(input_dim=2054, region_nums=object_params["object_num"], output_dim=256,
                                              time_module=object_params.get("time_module"),
                                              num_frames=object_params["num_frames"])
        if pretrained_init:
            load_clip_pt_weight(self.object_model)                                                # model/model.py:31-36
        if projection != "minimal":
            raise NotImplementedError(projection)
        # nn.Sequential(nn.ReLU(), nn.Linear(768, 256)): keep the index so the key is txt_proj.1.*
        self.txt_proj = nn.Sequential(nn.ReLU(), nn.Linear(self.text_model.config.hidden_size, projection_dim))
        self.set_compute_dtype(compute_dtype)
        if load_checkpoint not in ["", None]:
            checkpoint = torch.load(load_checkpoint, map_location="cpu")
            state_dict = checkpoint["state_dict"]
            new_state_dict = state_dict_data_parallel_fix(state_dict, self.state_dict())
            new_state_dict = self._inflate_positional_embeds(new_state_dict)
            try:
                self.load_state_dict(new_state_dict, strict=True)
            except Exception as e:  # same fallback as model/model.py:56-62
                print("Parameters of model and state_dict are mismatched. {}".format(e))
                self.load_state_dict_with_mismatch(new_state_dict)
        self.segments = object_params["num_frames"]
        self.projection_dim = 256
        # the two towers are independent until the loss: run the text tower on its own HIP stream so that its under-filled
        # launches (M = batch x 100 tokens) share the chip with the object tower's (default off: per-kernel timings stay clean)
        self.parallel_towers = False

    # ---- knobs that do not exist in the reference -------------------------------------------------------------
    def set_compute_dtype(self, dtype):
        self.compute_dtype = _DTYPES[dtype]
        self.text_model.compute_dtype = self.compute_dtype
        self.object_model.compute_dtype = self.compute_dtype
        return self

    def set_text_dropout(self, dropout, attention_dropout=None):
        """DistilBERT's train-mode dropout probabilities (default: the HF config's 0.1 / 0.1, as the reference trains)."""
        self.text_model.set_dropout(dropout, attention_dropout)
        return self

    # ---- reference surface ----------------------------------------------------------------------------------------
    def set_device(self, device):
        self.device = device

    def forward(self, data, return_embeds=True):
        text_data = data["text"]
        if self.parallel_towers and text_data["input_ids"].is_cuda:
            from . import ops
            main = torch.cuda.current_stream()
            ts = ops.text_stream()
            ts.wait_stream(main)
            with torch.cuda.stream(ts):
                gt, lt = self.compute_text(text_data)
            go, lo, object_mask = self.compute_object(data["object"], data["object_mask"])
            main.wait_stream(ts)
            gt.record_stream(main)
            lt.record_stream(main)
        else:
            gt, lt = self.compute_text(text_data)
            go, lo, object_mask = self.compute_object(data["object"], data["object_mask"])
        return dict(global_text_embeddings=gt.contiguous(), local_text_embeddings=lt.contiguous(),
                    global_object_embeddings=go.contiguous(), local_object_embeddings=lo.contiguous(),
                    object_mask=object_mask[:, 1:, ...].contiguous())

    def compute_text(self, text_data, pad=False):
        ids = text_data["input_ids"]
        if not ids.is_cuda:
            raise DemoVLPHipError("ObjectRelation runs on an MI355X device only (no CPU fallback): move the batch to cuda")
        h, hr = self.text_model(input_ids=ids, attention_mask=text_data.get("attention_mask"), want_relu=True)
        lin = self.txt_proj[1]
        emb = Fn.LinearFn.apply(hr, lin.weight, lin.bias, h)      # hr = relu(h); the gradient is routed to h
        return Fn.SplitClsFn.apply(emb)                            # emb[:, 0], emb[:, 1:] (:88-90), already contiguous

    def compute_object(self, object_data, object_mask):
        if not object_data.is_cuda:
            raise DemoVLPHipError("ObjectRelation runs on an MI355X device only (no CPU fallback): move the batch to cuda")
        emb, add_mask = self.object_model(object_data, object_mask)
        go, lo = Fn.SplitClsFn.apply(emb)                          # emb[:, 0], emb[:, 1:] (:94-96), already contiguous
        return go, lo, add_mask

    def _inflate_positional_embeds(self, new_state_dict):
        """model/model.py:98-151: adapt object_model.temporal_embed when the checkpoint's frame count differs."""
        key = "object_model.temporal_embed"
        curr = self.state_dict()
        if key in new_state_dict and key in curr:
            load = new_state_dict[key]
            n_load, n_curr, dim = load.shape[1], self.object_params["num_frames"], load.shape[2]
            if n_load > n_curr:
                new_state_dict[key] = load[:, :n_curr, :]
            elif n_load < n_curr:
                if self.load_temporal_fix == "zeros":
                    new = torch.zeros([load.shape[0], n_curr, dim])
                    new[:, :n_load] = load
                elif self.load_temporal_fix in ("interp", "bilinear"):
                    mode = "bilinear" if self.load_temporal_fix == "bilinear" else "nearest"
                    new = F.interpolate(load.unsqueeze(0), (n_curr, dim), mode=mode).squeeze(0)
                else:
                    raise NotImplementedError
                new_state_dict[key] = new
        key = "object_model.custom_pos_embed"
        if key in new_state_dict and key in curr and new_state_dict[key].shape[1] != curr[key].shape[1]:
            raise NotImplementedError("Loading models with different spatial resolution / patch number not yet implemented, sorry.")
        return new_state_dict

    def load_state_dict_with_mismatch(self, loaded_state_dict_or_path):
        """model/model.py:153-197: load every tensor whose name (with or without 'module.') and shape match."""
        loaded = torch.load(loaded_state_dict_or_path, map_location="cpu") if isinstance(loaded_state_dict_or_path, str) \
            else loaded_state_dict_or_path
        mine = self.state_dict()
        toload, mismatched, missing = {}, [], []
        for k, v in mine.items():
            src = loaded.get(k, loaded.get("module." + k))
            if src is None:
                missing.append(k)
            elif src.shape != v.shape:
                mismatched.append(k)
            else:
                toload[k] = src
        print(f"Keys in model but not in loaded: In total {len(missing)}, {sorted(missing)}")
        print(f"Keys in model and loaded, but shape mismatched: In total {len(mismatched)}, {sorted(mismatched)}")
        self.load_state_dict(toload, strict=False)


class ObjectQARelation(ObjectRelation):
    """Video question answering: the two towers + `BUTDQAHead` (mirror of model/model.py:200-289).  ``object_params`` carries
    ``num_label`` (configs/ft/msrvtt_qa-select.json:15).  ``forward(data) -> {'logits': [B, num_label]}``; the text embedding is
    the max over ALL 100 token projections (:285, padded tokens included, as the reference), the object side the local region
    embeddings with the loader's 0/1 region mask."""

    def __init__(self, object_params, text_params, projection_dim=256, load_checkpoint=None, projection="minimal",
                 load_temporal_fix="bilinear", compute_dtype="float32", pretrained_init=True):
        super().__init__(object_params, text_params, projection_dim, None, projection, load_temporal_fix, compute_dtype, pretrained_init)
        from .video_qa_model import BUTDQAHead
        self.head = BUTDQAHead(v_dim=256, q_dim=256, hid_dim=256, out_dim=object_params["num_label"])
        if load_checkpoint not in ["", None]:
            checkpoint = torch.load(load_checkpoint, map_location="cpu")
            new_state_dict = self._inflate_positional_embeds(state_dict_data_parallel_fix(checkpoint["state_dict"], self.state_dict()))
            try:
                self.load_state_dict(new_state_dict, strict=True)
            except Exception as e:
                print("Parameters of model and state_dict are mismatched. {}".format(e))
                self.load_state_dict_with_mismatch(new_state_dict)

    def forward(self, data, return_embeds=True):
        text_data = data["text"]
        ids = text_data["input_ids"]
        if not ids.is_cuda:
            raise DemoVLPHipError("ObjectQARelation runs on an MI355X device only (no CPU fallback): move the batch to cuda")
        h, hr = self.text_model(input_ids=ids, attention_mask=text_data.get("attention_mask"), want_relu=True)
        lin = self.txt_proj[1]
        text_embeddings = Fn.LinearFn.apply(hr, lin.weight, lin.bias, h)              # [B, 100, 256]
        object_embeddings, _ = self.object_model(data["object"], data["object_mask"])
        return dict(logits=self.compute_fusion(text_embeddings, object_embeddings, text_data.get("attention_mask"), data["object_mask"]))

    def compute_fusion(self, text_embeddings, object_embeddings, text_mask, object_mask):
        if object_mask.dim() == 3:
            object_mask = object_mask.reshape(object_mask.shape[0], -1)
        object_mask = object_mask.to(torch.float32)
        text_embeddings, _ = torch.max(text_embeddings.float(), dim=1)
        return self.head(text_embeddings, object_embeddings[:, 1:].float(), object_mask)


class ObjectMCRelation(ObjectRelation):
    """Multiple-choice evaluation model (model/model.py:393-579): the same two towers, `forward` and state_dict as ObjectRelation
    (the reference class is a copy of it; only the trainer that consumes the embeddings differs)."""

    def __init__(self, object_params, text_params, projection_dim=256, load_checkpoint=None, projection="minimal",
                 load_temporal_fix="zeros", compute_dtype="float32", pretrained_init=True):
        super().__init__(object_params, text_params, projection_dim, load_checkpoint, projection, load_temporal_fix, compute_dtype, pretrained_init)


def sim_matrix(a, b, eps=1e-8):
    """model/model.py:582-590 for any [N,256] x [M,256].  Rows are l2-normalised with max(|.|, 1e-8), then a_n b_n^T.  Host
    tensors (the reference's validation path hands over .cpu() tensors, trainer_dist.py:369) are staged through the GPU and the
    result comes back on the host; without a GPU this raises."""
    if eps != 1e-8:
        raise NotImplementedError("the kernels hard-code eps = 1e-8 (the only value the reference uses)")
    if a.is_cuda:
        return Fn.SimMatrixFn.apply(a, b)
    if not torch.cuda.is_available():
        raise DemoVLPHipError("sim_matrix needs an MI355X device (no CPU fallback)")
    return Fn.SimMatrixFn.apply(a.cuda().float(), b.cuda().float()).cpu()
