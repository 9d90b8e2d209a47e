"""Deterministic synthetic inputs and weights for the DemoVLP hot path.

numpy-only (no torch compute, no GPU).  Used by the golden-fixture generator, the tests, ``bench.py``
and ``__graft_entry__.smoke()`` so that the reference model (in the build container), the CPU oracle and
the HIP path all see bit-identical inputs and parameters without shipping 600 MB of weights.

Schemas follow what the reference reads:
  * per-frame region file: keys ``x [Nraw,2048] f32``, ``bbox [Nraw,4] f32``, ``info`` (0-d object array
    holding ``{objects_conf, objects_id, image_w, image_h}``) -- data_loader/WebVid_dataset.py:243-256
  * captions: already-tokenised ``input_ids`` / ``attention_mask`` ``[B,100] int64`` -- the trainer pads
    every caption to ``max_length=100`` (trainer/trainer_dist.py:132-137)
"""
from __future__ import annotations

import os
import zlib

import numpy as np

FEAT_DIM = 2048
BOX_DIM = 6
TEXT_LEN = 100
VOCAB = 30522


# --------------------------------------------------------------------------------------------------
# region files
# --------------------------------------------------------------------------------------------------
def make_frame(sample: int, frame: int, n_raw: int = 36, image_w: int = 640, image_h: int = 360) -> dict:
    """One synthetic bottom-up-attention frame."""
    rng = np.random.default_rng(1234 + 64 * sample + frame)
    x = np.maximum(rng.standard_normal((n_raw, FEAT_DIM), dtype=np.float32), 0.0).astype(np.float32)
    x0 = (rng.random(n_raw, dtype=np.float32) * np.float32(0.6) * np.float32(image_w)).astype(np.float32)
    y0 = (rng.random(n_raw, dtype=np.float32) * np.float32(0.6) * np.float32(image_h)).astype(np.float32)
    bw = ((np.float32(0.05) + rng.random(n_raw, dtype=np.float32) * np.float32(0.35)) * np.float32(image_w)).astype(np.float32)
    bh = ((np.float32(0.05) + rng.random(n_raw, dtype=np.float32) * np.float32(0.35)) * np.float32(image_h)).astype(np.float32)
    bbox = np.stack([x0, y0, x0 + bw, y0 + bh], axis=1).astype(np.float32)
    while True:
        conf = (np.float32(0.2) + rng.random(n_raw, dtype=np.float32) * np.float32(0.8)).astype(np.float32)
        if len(np.unique(conf)) == n_raw:  # ties are order-undefined in numpy's argsort -> keep distinct
            break
    ids = rng.integers(0, 1600, n_raw).astype(np.int64)
    return dict(x=x, bbox=bbox, objects_conf=conf, objects_id=ids, image_w=image_w, image_h=image_h)


def save_frame_npz(path: str, frame: dict) -> None:
    """Write a frame in the exact .npz schema the reference loader expects."""
    info = np.array(dict(objects_conf=frame["objects_conf"], objects_id=frame["objects_id"],
                         image_w=frame["image_w"], image_h=frame["image_h"]), dtype=object)
    np.savez(path, x=frame["x"], bbox=frame["bbox"], info=info)


def raw_region_batch(batch: int, frames: int, n_raw: int = 36, first_sample: int = 0):
    """Raw (unselected) region tensors for a batch: feats [B,F,Nraw,2048], bbox [B,F,Nraw,4],
    conf [B,F,Nraw], image_wh [B,F,2] (all f32)."""
    feats = np.empty((batch, frames, n_raw, FEAT_DIM), np.float32)
    bbox = np.empty((batch, frames, n_raw, 4), np.float32)
    conf = np.empty((batch, frames, n_raw), np.float32)
    wh = np.empty((batch, frames, 2), np.float32)
    for b in range(batch):
        for f in range(frames):
            fr = make_frame(first_sample + b, f, n_raw)
            feats[b, f], bbox[b, f], conf[b, f] = fr["x"], fr["bbox"], fr["objects_conf"]
            wh[b, f] = (fr["image_w"], fr["image_h"])
    return feats, bbox, conf, wh


def fast_region_batch(batch: int, frames: int, regions: int, seed: int = 7, pad_every: int = 5):
    """Cheap already-selected batch for throughput runs: object [B,F,R,2054] f32, mask [B,F,R] f32.
    Every ``pad_every``-th frame has 3 padded (masked) regions so the mask path is exercised."""
    rng = np.random.default_rng(seed)
    obj = np.maximum(rng.standard_normal((batch, frames, regions, FEAT_DIM + BOX_DIM), dtype=np.float32), 0.0)
    box = rng.random((batch, frames, regions, BOX_DIM), dtype=np.float32)
    obj[..., FEAT_DIM:] = box
    mask = np.ones((batch, frames, regions), np.float32)
    flat = mask.reshape(-1, regions)
    flat[::pad_every, regions - 3:] = 0.0
    return obj.astype(np.float32), mask


# --------------------------------------------------------------------------------------------------
# captions
# --------------------------------------------------------------------------------------------------
def caption_batch(batch: int, first_sample: int = 0, text_len: int = TEXT_LEN):
    """input_ids / attention_mask [B,text_len] int64: [CLS]=101 + random word pieces + [SEP]=102, 0-padded."""
    ids = np.zeros((batch, text_len), np.int64)
    att = np.zeros((batch, text_len), np.int64)
    for b in range(batch):
        rng = np.random.default_rng(4321 + first_sample + b)
        n = int(rng.integers(6, 33))
        body = rng.integers(1000, VOCAB, n - 2)
        ids[b, 0] = 101
        ids[b, 1:n - 1] = body
        ids[b, n - 1] = 102
        att[b, :n] = 1
    return ids, att


# --------------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------------
def qa_head_schema(num_label: int) -> "dict[str, tuple]":
    """`head.*` tensors of ObjectQARelation (BUTDQAHead, model/video_qa_mdoel.py:78-97; weight-normalised linears carry weight_g / weight_v)."""
    s: "dict[str, tuple]" = {}
    for p in ("head.v_att.v_proj.main.0", "head.v_att.q_proj.main.0", "head.classifier.q_net.main.0", "head.classifier.v_net.main.0"):
        s[p + ".bias"] = (256,)
        s[p + ".weight_g"] = ()
        s[p + ".weight_v"] = (256, 256)
    s["head.v_att.linear.bias"] = (1,)
    s["head.v_att.linear.weight_g"] = ()
    s["head.v_att.linear.weight_v"] = (1, 256)
    s["head.classifier.main.0.weight"] = (512, 256)
    s["head.classifier.main.0.bias"] = (512,)
    s["head.classifier.main.3.weight"] = (num_label, 512)
    s["head.classifier.main.3.bias"] = (num_label,)
    return s


def state_dict_schema(num_frames: int, object_num: int, time_module=None) -> "dict[str, tuple]":
    """Name -> shape of every tensor in ``ObjectRelation.state_dict()`` (280 tensors, + 4 per block
    with ``time_module='timeattn'``: model/object_transformer.py:227-234)."""
    D, Hd = 768, 3072
    s: "dict[str, tuple]" = {}
    s["text_model.embeddings.word_embeddings.weight"] = (VOCAB, D)
    s["text_model.embeddings.position_embeddings.weight"] = (512, D)
    s["text_model.embeddings.LayerNorm.weight"] = (D,)
    s["text_model.embeddings.LayerNorm.bias"] = (D,)
    for l in range(6):
        p = f"text_model.transformer.layer.{l}."
        for n in ("q_lin", "k_lin", "v_lin", "out_lin"):
            s[p + f"attention.{n}.weight"] = (D, D)
            s[p + f"attention.{n}.bias"] = (D,)
        s[p + "sa_layer_norm.weight"] = (D,)
        s[p + "sa_layer_norm.bias"] = (D,)
        s[p + "ffn.lin1.weight"] = (Hd, D)
        s[p + "ffn.lin1.bias"] = (Hd,)
        s[p + "ffn.lin2.weight"] = (D, Hd)
        s[p + "ffn.lin2.bias"] = (D,)
        s[p + "output_layer_norm.weight"] = (D,)
        s[p + "output_layer_norm.bias"] = (D,)
    s["object_model.cls_token"] = (1, 1, D)
    s["object_model.custom_pos_embed"] = (1, object_num + 1, D)
    s["object_model.temporal_embed"] = (1, num_frames, D)
    for l in range(12):
        p = f"object_model.blocks.{l}."
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        if time_module == "timeattn":
            s[p + "timeattn.qkv.weight"] = (3 * D, D)
            s[p + "timeattn.qkv.bias"] = (3 * D,)
            s[p + "timeattn.proj.weight"] = (D, D)
            s[p + "timeattn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (Hd, D)
        s[p + "mlp.fc1.bias"] = (Hd,)
        s[p + "mlp.fc2.weight"] = (D, Hd)
        s[p + "mlp.fc2.bias"] = (D,)
        s[p + "norm3.weight"] = (D,)
        s[p + "norm3.bias"] = (D,)
    s["object_model.norm.weight"] = (D,)
    s["object_model.norm.bias"] = (D,)
    s["object_model.object_embedding.weight"] = (D, FEAT_DIM)
    s["object_model.object_embedding.bias"] = (D,)
    s["object_model.pos_embedding.weight"] = (D, BOX_DIM)
    s["object_model.pos_embedding.bias"] = (D,)
    s["object_model.proj.weight"] = (256, D)
    s["txt_proj.1.weight"] = (256, D)
    s["txt_proj.1.bias"] = (256,)
    return s


def fill_tensor(name: str, shape) -> np.ndarray:
    """Deterministic value of one parameter: a PCG64 stream keyed by crc32(name).

    LayerNorm gains sit around 1, biases are small, matrices use std 0.02 (ViT/BERT init scale) except the
    two 256-d projection heads which get a larger std so the contrastive logits are not degenerate."""
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    shape = tuple(shape)
    z = rng.standard_normal(shape, dtype=np.float32)
    leaf = name.rsplit(".", 1)[-1]
    is_norm = ("norm" in name.lower()) and len(shape) == 1
    if is_norm and leaf == "weight":
        return (1.0 + 0.1 * z).astype(np.float32)
    if leaf == "weight_g":
        return np.asarray(4.0 + 0.5 * z, dtype=np.float32)  # weight-norm gain (a scalar: dim=None)
    if leaf == "bias":
        return (0.02 * z).astype(np.float32)
    if name in ("object_model.proj.weight", "txt_proj.1.weight"):
        return (0.05 * z).astype(np.float32)
    if name == "object_model.pos_embedding.weight":
        return (0.2 * z).astype(np.float32)
    return (0.02 * z).astype(np.float32)


def vit_checkpoint_schema(blocks=range(12)) -> "dict[str, tuple]":
    """Name -> shape of timm's ``vit_base_patch16_224`` state_dict (the file ``pretrained/jx_vit_base_p16_224-80ecf9dd.pth`` that
    ``load_clip_pt_weight`` reads, model/object_transformer.py:470-483), restricted to ``blocks``.  Of these, ``pos_embed``,
    ``patch_embed.*`` and ``head.*`` have no counterpart in ObjectTransformer (ignored by strict=False)."""
    D, Hd = 768, 3072
    s: "dict[str, tuple]" = {"cls_token": (1, 1, D), "pos_embed": (1, 197, D), "patch_embed.proj.weight": (D, 3, 16, 16), "patch_embed.proj.bias": (D,)}
    for l in blocks:
        p = f"blocks.{l}."
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (Hd, D)
        s[p + "mlp.fc1.bias"] = (Hd,)
        s[p + "mlp.fc2.weight"] = (D, Hd)
        s[p + "mlp.fc2.bias"] = (D,)
    s["norm.weight"] = (D,)
    s["norm.bias"] = (D,)
    s["head.weight"] = (1000, D)
    s["head.bias"] = (1000,)
    return s


def vit_checkpoint(blocks=range(12)) -> "dict[str, np.ndarray]":
    """A synthetic stand-in for the timm checkpoint: closed-form values keyed ``vit/<name>`` (distinct from every model fill)."""
    return {k: fill_tensor("vit/" + k, shp) for k, shp in vit_checkpoint_schema(blocks).items()}


_FILL_CACHE: "dict[tuple, dict]" = {}
_FILL_CACHE_MAX = 3          # 614 MB per full-size entry; the test suites rebuild the same two or three models ~100 times (4.3 s of PCG64 each)


def fill_state_dict(num_frames: int, object_num: int, time_module=None, qa_labels: int = 0) -> "dict[str, np.ndarray]":
    """The closed-form weights of one model: a pure function of the arguments (every tensor a PCG64 stream keyed by its name).  The last
    few results are kept and handed out as COPIES (0.2 s instead of 4.3 s; callers may edit what they get)."""
    key = (int(num_frames), int(object_num), time_module or None, int(qa_labels))
    hit = _FILL_CACHE.get(key)
    if hit is None:
        schema = state_dict_schema(num_frames, object_num, time_module)
        if qa_labels:
            schema.update(qa_head_schema(qa_labels))
        # (280 independent PCG64 streams; numpy's generators release the GIL: a few threads take the 4 s of a full-size model to ~1 s)
        from concurrent.futures import ThreadPoolExecutor
        names = list(schema)
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            vals = list(ex.map(lambda k: fill_tensor(k, schema[k]), names))
        hit = dict(zip(names, vals))
        while len(_FILL_CACHE) >= _FILL_CACHE_MAX:
            _FILL_CACHE.pop(next(iter(_FILL_CACHE)))
        _FILL_CACHE[key] = hit
    return {k: v.copy() for k, v in hit.items()}


# --------------------------------------------------------------------------------------------------
# a retrieval set with SIGNAL (golden G11): weights and inputs built so that video i and caption i score above chance
# --------------------------------------------------------------------------------------------------
# With the closed-form random weights above the two towers are unrelated, every eval set retrieves at chance (G9: R@1 = 1/256) and a
# rank swap changes nothing visible.  G11 needs matched pairs to win -- by a margin small enough that R@1 sits near one half, where
# every rank swap shows.  Everything stays a pure function of seeds:
#   weights  the closed-form fill, then (a) every residual branch's output projection scaled by RETR_BRANCH (both towers pass their
#            embedded input through nearly unchanged), (b) object_embedding.weight = [I_768 | 0] (the first 768 feature channels ARE
#            the embedded token), (c) txt_proj.1.weight = object_model.proj.weight (one shared 256-d head);
#   captions groups of RETR_GROUP captions share RETR_SHARED of their RETR_WORDS body tokens (hard negatives);
#   regions  region (f, r) of clip i aims at one body word w of caption i: its feature vector is chosen so that the EMBEDDED token equals
#            relu(LN_emb(word_emb[id_w] + pos_emb[w])) (the text tower's embedding output, which its six damped layers barely move)
#            plus Gaussian noise of RETR_NOISE times its rms -- i.e. feature[:768] = target - bias - box term - temporal embedding.
RETR_BRANCH, RETR_GROUP, RETR_WORDS, RETR_SHARED, RETR_NOISE = 0.05, 8, 12, 11, 6.0


def retrieval_state_dict(num_frames: int, object_num: int) -> "dict[str, np.ndarray]":
    sd = fill_state_dict(num_frames, object_num)
    for k in list(sd):
        if k.endswith(("attn.proj.weight", "attn.proj.bias", "mlp.fc2.weight", "mlp.fc2.bias", "attention.out_lin.weight", "attention.out_lin.bias",
                       "ffn.lin2.weight", "ffn.lin2.bias")):
            sd[k] = (sd[k] * np.float32(RETR_BRANCH)).astype(np.float32)
    w = np.zeros((768, FEAT_DIM), np.float32)
    w[np.arange(768), np.arange(768)] = 1.0
    sd["object_model.object_embedding.weight"] = w
    sd["txt_proj.1.weight"] = sd["object_model.proj.weight"].copy()
    return sd


def retrieval_captions(first: int, batch: int, text_len: int = TEXT_LEN):
    ids = np.zeros((batch, text_len), np.int64)
    att = np.zeros((batch, text_len), np.int64)
    for b in range(batch):
        i = first + b
        shared = np.random.default_rng(777 + i // RETR_GROUP).integers(1000, VOCAB, RETR_SHARED)
        own = np.random.default_rng(99991 + i).integers(1000, VOCAB, RETR_WORDS - RETR_SHARED)
        body = np.concatenate([shared, own])
        body = body[np.random.default_rng(5 + i).permutation(RETR_WORDS)]
        n = RETR_WORDS + 2
        ids[b, 0], ids[b, 1:n - 1], ids[b, n - 1] = 101, body, 102
        att[b, :n] = 1
    return ids, att


def retrieval_batch(sd, num_frames: int, object_num: int, first: int, batch: int):
    """(object [B,F,R,2054] f32, mask [B,F,R] f32, input_ids, attention_mask) of pairs first .. first + batch - 1 for the weights ``sd``
    (= retrieval_state_dict).  float64 inside, rounded once: the same bytes wherever numpy runs."""
    F_, R = num_frames, object_num
    ids, att = retrieval_captions(first, batch)
    we = sd["text_model.embeddings.word_embeddings.weight"].astype(np.float64)
    pe = sd["text_model.embeddings.position_embeddings.weight"].astype(np.float64)
    g, bt = sd["text_model.embeddings.LayerNorm.weight"].astype(np.float64), sd["text_model.embeddings.LayerNorm.bias"].astype(np.float64)
    be = sd["object_model.object_embedding.bias"].astype(np.float64)
    wp, bp = sd["object_model.pos_embedding.weight"].astype(np.float64), sd["object_model.pos_embedding.bias"].astype(np.float64)
    te = sd["object_model.temporal_embed"].astype(np.float64)[0]
    obj = np.zeros((batch, F_, R, FEAT_DIM + BOX_DIM), np.float32)
    mask = np.ones((batch, F_, R), np.float32)
    for b in range(batch):
        i = first + b
        rng = np.random.default_rng(424242 + i)
        pos = 1 + rng.integers(0, RETR_WORDS, (F_, R))                       # the body word each region aims at
        x = we[ids[b, pos]] + pe[pos]
        x = (x - x.mean(-1, keepdims=True)) / np.sqrt(x.var(-1, keepdims=True) + 1e-12) * g + bt
        tgt = np.maximum(x, 0.0)
        tgt = tgt + RETR_NOISE * np.sqrt((tgt ** 2).mean()) * rng.standard_normal(tgt.shape)
        box = rng.random((F_, R, BOX_DIM))
        feat = rng.standard_normal((F_, R, FEAT_DIM)) * 0.5                   # channels >= 768 are ignored by the [I | 0] embedding
        feat[..., :768] = tgt - be - box @ wp.T - bp - te[:F_, None, :]
        obj[b, ..., :FEAT_DIM], obj[b, ..., FEAT_DIM:] = feat, box
        npad = i % 4
        if npad:
            mask[b, F_ - 1, R - npad:] = 0.0                                   # a few padded regions: the mask path stays exercised
    return obj, mask, ids, att
