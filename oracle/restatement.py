"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A plain-PyTorch fp32 (numpy for the integer top-k) restatement of the DemoVLP cross-modal hot path, written
from the maths in SURVEY.md section 8(a), not from the reference's code structure.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this module; the shipped
package ``demovlp_amd`` never does, and fails loudly when its HIP library is missing.

Parity pinning: every function here is checked against golden vectors produced by importing the *unmodified*
reference in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; test:
``tests/test_oracle_golden.py``).  Third-party arithmetic (HuggingFace DistilBERT 4.10.0, transformers.AdamW
4.10.0) is not under /root/reference: DistilBERT is pinned by goldens from the container's transformers
5.15.0; HF-AdamW is restated from its documented update and is "parity unpinned".

All reference citations are paths under /root/reference.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

FEAT_DIM = 2048
EMBED = 768
HEADS = 12
HEAD_DIM = 64


# ==================================================================================================
# A1  region select (integer/index work -> numpy, bit-exact target)
# ==================================================================================================
def region_select(feats, bbox, conf, image_w, image_h, object_num):
    """One clip.  feats: list over frames of [Nraw_f,2048] f32; bbox: list of [Nraw_f,4]; conf: list of [Nraw_f].

    Follows data_loader/WebVid_dataset.py:231-283 (sort by confidence descending, 6-d geometry) and
    :151-228 (keep the first ``object_num`` rows, else edge-pad with the last row; 0/1 mask).
    Returns object [F,R,2054] f32, mask [F,R] f64, lengths list[int], order list[np.ndarray int64].
    """
    F_ = len(feats)
    R = int(object_num)
    out = np.zeros((F_, R, FEAT_DIM + 6), np.float32)
    mask = np.zeros((F_, R), np.float64)
    lens, orders = [], []
    for f in range(F_):
        order = np.argsort(conf[f])[::-1]                       # WebVid_dataset.py:249
        x = feats[f][order]
        b = bbox[f][order]
        w = b[:, 2] - b[:, 0]
        h = b[:, 3] - b[:, 1]
        sw, sh = w / image_w, h / image_h                      # :257-262
        sx, sy = b[:, 0] / image_w, b[:, 1] / image_h
        geo = np.stack([sx, sy, sx + sw, sy + sh, sw, sh], axis=1).astype(np.float32)   # :267-270
        n = x.shape[0]
        keep = min(n, R)                                        # :192-207
        row = np.concatenate([x[:keep], geo[:keep]], axis=1)
        out[f, :keep] = row
        if keep < R:                                            # np.pad(..., 'edge') :208-214
            out[f, keep:] = row[keep - 1]
        mask[f, :keep] = 1.0                                    # :219-221
        lens.append(keep)
        orders.append(order[:keep].astype(np.int64))
    return out, mask, lens, orders


# ==================================================================================================
# A2-A6  object transformer
# ==================================================================================================
def structural_attention_mask(F_, R, device=None, time=False):
    """Boolean [N,N] (N=1+F*R): query i may see key j iff i is CLS, or j is CLS, or same frame (space attention,
    'b (f n) d -> (b f) n d') / same region slot (time attention, 'b (f n) d -> (b n) f d').

    Equivalent form of VarAttention's CLS-splice + rearrange (model/object_transformer.py:162-189)."""
    N = 1 + F_ * R
    frame = torch.full((N,), -1, dtype=torch.long, device=device)
    frame[1:] = (torch.arange(F_ * R, device=device) % R) if time else (torch.arange(F_ * R, device=device) // R)
    same = frame[:, None] == frame[None, :]
    allow = same | (frame[None, :] == -1) | (frame[:, None] == -1)
    return allow


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def object_prologue(p, obj, mask01):
    """A2: model/object_transformer.py:400-433.  obj [B,F,R,2054], mask01 [B,F,R] (1 = real region).
    Returns x [B,N,768] and the additive key mask [B,N] (0 / -100, CLS = 0)."""
    B, F_, R, _ = obj.shape
    pre = "object_model."
    feat, box = obj[..., :FEAT_DIM], obj[..., FEAT_DIM:]
    tok = feat @ p[pre + "object_embedding.weight"].t() + p[pre + "object_embedding.bias"]
    tok = tok + box @ p[pre + "pos_embedding.weight"].t() + p[pre + "pos_embedding.bias"]
    tok = tok.reshape(B, F_ * R, EMBED)
    tok = tok + p[pre + "temporal_embed"][0, :F_].repeat_interleave(R, dim=0)[None]  # :425-432 (curr_frames <= num_frames: first F rows)
    cls = (p[pre + "cls_token"][0, 0] + p[pre + "custom_pos_embed"][0, 0])[None, None].expand(B, 1, EMBED)
    x = torch.cat([cls, tok], dim=1)
    m = torch.cat([torch.ones(B, 1, dtype=mask01.dtype, device=mask01.device), mask01.reshape(B, -1)], dim=1)
    return x, (m - 1.0) * 100.0                                                       # :421


def space_attention(qkv, add_mask, F_, R, time=False):
    """A4: qkv [B,N,2304] -> [B,N,768].  Full attention under the structural mask plus additive key mask."""
    B, N, _ = qkv.shape
    q, k, v = qkv.reshape(B, N, 3, HEADS, HEAD_DIM).permute(2, 0, 3, 1, 4)           # [B,H,N,64]
    s = (q * HEAD_DIM ** -0.5) @ k.transpose(-1, -2)                                  # :160
    s = s + add_mask[:, None, None, :].to(s.dtype)
    allow = structural_attention_mask(F_, R, qkv.device, time)
    s = s.masked_fill(~allow[None, None], float("-inf"))
    a = torch.softmax(s, dim=-1)
    return (a @ v).transpose(1, 2).reshape(B, N, EMBED)


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def vit_block(p, pre, x, add_mask, F_, R):
    """A3/A5: SpaceTimeBlock (model/object_transformer.py:249-274).  With the 'timeattn' weights present (time_module='timeattn',
    :252-258): time attention over norm3(x) first, space attention on norm1(x + time), and -- the FrozenInTime residual --
    the space output is added to the block INPUT x, not to the time residual (:267)."""
    xin = x
    if pre + "timeattn.qkv.weight" in p:
        h3 = layer_norm(x, p[pre + "norm3.weight"], p[pre + "norm3.bias"], 1e-6)
        tq = h3 @ p[pre + "timeattn.qkv.weight"].t() + p[pre + "timeattn.qkv.bias"]
        ta = space_attention(tq, add_mask, F_, R, time=True)
        xin = x + ta @ p[pre + "timeattn.proj.weight"].t() + p[pre + "timeattn.proj.bias"]
    h = layer_norm(xin, p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-6)
    qkv = h @ p[pre + "attn.qkv.weight"].t() + p[pre + "attn.qkv.bias"]
    a = space_attention(qkv, add_mask, F_, R)
    x1 = x + a @ p[pre + "attn.proj.weight"].t() + p[pre + "attn.proj.bias"]
    h2 = layer_norm(x1, p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-6)
    m = gelu_erf(h2 @ p[pre + "mlp.fc1.weight"].t() + p[pre + "mlp.fc1.bias"])
    return x1 + m @ p[pre + "mlp.fc2.weight"].t() + p[pre + "mlp.fc2.bias"]


def object_encoder(p, obj, mask01, taps=None):
    """A2-A6.  Returns (embeddings [B,N,256], additive mask [B,N]).  No final LayerNorm (:449-452)."""
    B, F_, R, _ = obj.shape
    x, add_mask = object_prologue(p, obj, mask01)
    if taps is not None:
        taps["embed"] = x
    for l in range(12):
        x = vit_block(p, f"object_model.blocks.{l}.", x, add_mask, F_, R)
        if taps is not None:
            taps[f"block{l}"] = x
    return x @ p["object_model.proj.weight"].t(), add_mask


# ==================================================================================================
# A8  DistilBERT + txt_proj   (third-party arithmetic, see module docstring)
# ==================================================================================================
# Philox4x32-10 (Salmon et al., SC'11), numpy: the counter-based generator behind the HIP path's dropout masks
# (demovlp_amd/csrc/common.h).  Pinned by the published Random123 known-answer vectors (tests/test_oracle_golden.py).
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c = [np.asarray(x, np.uint64) for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32, p0 & m32]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m32, (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def _drop_threshold(p):
    return np.uint32(min(int(float(np.float32(p)) * 4294967296.0), 0xFFFFFFFF))


def dropout_keep_flat(n, p, seed, offset, site):
    """keep mask (float 0/1) of a flat tensor of n elements (n % 4 == 0): element e uses word e % 4 of Philox(counter
    (e // 4, 0, site, offset), key (seed_lo, seed_hi)) -- csrc/dropout.hip:dropout_fwd_kernel."""
    i = np.arange(n // 4, dtype=np.uint64)
    u = philox4x32_10(i & np.uint64(0xFFFFFFFF), i >> np.uint64(32), site, offset, seed & 0xFFFFFFFF, seed >> 32)
    return (u.reshape(-1) >= _drop_threshold(p)).astype(np.float32)


def dropout_keep_attention(B, L, p, seed, offset, site):
    """keep mask [B, H, L, L] of the attention probabilities: row (b H + h) L + q, key group j // 4 -> counter
    (row, j // 4, site, offset) -- csrc/dropout.hip:dropout_attn_mask_kernel."""
    ng = (L + 3) // 4
    row = np.arange(B * HEADS * L, dtype=np.uint64)[:, None]
    jg = np.arange(ng, dtype=np.uint64)[None, :]
    u = philox4x32_10(row, jg, site, offset, seed & 0xFFFFFFFF, seed >> 32)            # [rows, ng, 4]
    keep = (u.reshape(B * HEADS * L, ng * 4)[:, :L] >= _drop_threshold(p)).astype(np.float32)
    return keep.reshape(B, HEADS, L, L)


def text_attention(q, k, v, att_mask01, keep=None, p_drop=0.0):
    B, L, _ = q.shape
    sh = lambda t: t.reshape(B, L, HEADS, HEAD_DIM).transpose(1, 2)
    s = (sh(q) @ sh(k).transpose(-1, -2)) * HEAD_DIM ** -0.5
    s = s.masked_fill(att_mask01[:, None, None, :] == 0, float("-inf"))
    w = torch.softmax(s, dim=-1)
    if keep is not None:
        w = w * keep / (1.0 - p_drop)              # HF MultiHeadSelfAttention: weights = dropout(softmax(scores))
    return (w @ sh(v)).transpose(1, 2).reshape(B, L, EMBED)


def text_encoder(p, input_ids, att_mask01, taps=None, drop=None):
    """DistilBERT (6 post-LN layers, eps 1e-12) then txt_proj = ReLU -> Linear (model/model.py:39-43, 86-90).
    Returns [B,L,256].  ``drop`` = dict(p, p_attention, seed, offset): HF's train-mode dropouts (the reference keeps the text model
    in train mode, model/model.py:29-30) with the HIP path's Philox masks: embeddings (site 0), attention probabilities of
    layer l (site 1 + 2 l), feed-forward output of layer l (site 2 + 2 l)."""
    pre = "text_model."
    L = input_ids.shape[1]
    # nn.Embedding(vocab, dim, padding_idx=pad_token_id=0): row 0 never receives a gradient
    x = F.embedding(input_ids, p[pre + "embeddings.word_embeddings.weight"], padding_idx=0)
    x = x + p[pre + "embeddings.position_embeddings.weight"][:L][None]
    x = layer_norm(x, p[pre + "embeddings.LayerNorm.weight"], p[pre + "embeddings.LayerNorm.bias"], 1e-12)
    B = input_ids.shape[0]

    def flat_drop(t, site):
        if drop is None or drop["p"] <= 0.0:
            return t
        keep = torch.from_numpy(dropout_keep_flat(t.numel(), drop["p"], drop["seed"], drop["offset"], site)).reshape(t.shape).to(t.dtype)
        return t * keep / (1.0 - drop["p"])

    x = flat_drop(x, 0)
    for l in range(6):
        lp = pre + f"transformer.layer.{l}."
        lin = lambda t, n: t @ p[lp + n + ".weight"].t() + p[lp + n + ".bias"]
        keep = None
        if drop is not None and drop["p_attention"] > 0.0:
            keep = torch.from_numpy(dropout_keep_attention(B, L, drop["p_attention"], drop["seed"], drop["offset"], 1 + 2 * l)).to(x.dtype)
        a = text_attention(lin(x, "attention.q_lin"), lin(x, "attention.k_lin"), lin(x, "attention.v_lin"), att_mask01, keep,
                           drop["p_attention"] if drop is not None else 0.0)
        x = layer_norm(lin(a, "attention.out_lin") + x, p[lp + "sa_layer_norm.weight"], p[lp + "sa_layer_norm.bias"], 1e-12)
        f = flat_drop(lin(gelu_erf(lin(x, "ffn.lin1")), "ffn.lin2"), 2 + 2 * l)
        x = layer_norm(f + x, p[lp + "output_layer_norm.weight"], p[lp + "output_layer_norm.bias"], 1e-12)
        if taps is not None:
            taps[f"text_layer{l}"] = x
    return torch.relu(x) @ p["txt_proj.1.weight"].t() + p["txt_proj.1.bias"]


def model_forward(p, input_ids, att_mask01, obj, mask01, drop=None):
    """A7: ObjectRelation.forward (model/model.py:70-96)."""
    t = text_encoder(p, input_ids, att_mask01, drop=drop)
    o, add_mask = object_encoder(p, obj, mask01)
    return dict(global_text_embeddings=t[:, 0].contiguous(), local_text_embeddings=t[:, 1:].contiguous(),
                global_object_embeddings=o[:, 0].contiguous(), local_object_embeddings=o[:, 1:].contiguous(),
                object_mask=add_mask[:, 1:].contiguous())


# ==================================================================================================
# (f4)  QA head: ObjectQARelation.forward + BUTDQAHead (model/model.py:260-289, model/video_qa_mdoel.py:58-97)
# ==================================================================================================
def _wn_linear(p, pre, x):
    """weight_norm(nn.Linear, dim=None): w = g * v / |v|_F (one scalar gain for the whole matrix)."""
    v = p[pre + ".weight_v"]
    return x @ (p[pre + ".weight_g"] * v / v.norm()).t() + p[pre + ".bias"]


def qa_logits(p, input_ids, att_mask01, obj, mask01):
    t = text_encoder(p, input_ids, att_mask01)                         # [B, 100, 256]
    o, _ = object_encoder(p, obj, mask01)
    q = t.max(dim=1).values                                            # max over all tokens, padded ones included (model.py:285)
    v = o[:, 1:]
    m = mask01.reshape(mask01.shape[0], -1).to(v.dtype)
    fc = lambda pre, x: torch.relu(_wn_linear(p, pre + ".main.0", x))
    logits = _wn_linear(p, "head.v_att.linear", fc("head.v_att.v_proj", v) * fc("head.v_att.q_proj", q)[:, None]) * m[..., None]
    att = torch.softmax(logits, dim=1)                                 # padded regions keep logit 0 (mask is multiplied, not added)
    joint = fc("head.classifier.q_net", q) * fc("head.classifier.v_net", (att * v).sum(1))
    h = torch.relu(joint @ p["head.classifier.main.0.weight"].t() + p["head.classifier.main.0.bias"])
    return h @ p["head.classifier.main.3.weight"].t() + p["head.classifier.main.3.bias"]


# ==================================================================================================
# A9-A11  losses
# ==================================================================================================
def sim_matrix(a, b, eps=1e-8):
    """model/model.py:582-590."""
    an = a / a.norm(dim=1, keepdim=True).clamp_min(eps)
    bn = b / b.norm(dim=1, keepdim=True).clamp_min(eps)
    return an @ bn.t()


def norm_softmax_loss(x, temperature=0.05):
    """model/loss.py:126-138."""
    i = torch.diagonal(torch.log_softmax(x / temperature, dim=1)).mean()
    j = torch.diagonal(torch.log_softmax(x.t() / temperature, dim=1)).mean()
    return -i - j


def _unit(x, eps=1e-8):
    return x / (x.pow(2).sum(-1, keepdim=True).sqrt() + eps)              # model/loss.py:333-338


def _cos_rows(a, b, eps=1e-8):
    return (a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1)).clamp_min(eps)   # model/loss.py:286-291


def _focal_softmax(logits, gate):
    """softmax over the last axis, 'equal' focal gate, renormalise (model/loss.py:245-259, 274-283).
    The gate is a constant (torch.where on a comparison) for autograd."""
    P = torch.softmax(logits, dim=-1)
    if gate:
        L = P.shape[-1]
        H = ((P * L - P.sum(-1, keepdim=True)) > 0).to(P.dtype)
        P = H * P
    return P / P.sum(-1, keepdim=True)


def xattn_pair(C, Q, m_img, m_cap, lam=20.0, gate=True):
    """Score of ONE (video i, caption j) pair.  C [G,d] regions, Q [W,d] words (raw), additive masks [G], [W].
    model/loss.py:294-330 with :209-271 unrolled for a single pair."""
    Ch, Qh = _unit(C), _unit(Q)
    S = F.leaky_relu(Ch @ Qh.t(), 0.1)                                     # [G,W]  :235-236
    # image -> text: normalise each region over words, softmax over regions for each word
    A = S / (S.pow(2).sum(1, keepdim=True).sqrt() + 1e-8)                  # :238
    P = _focal_softmax(lam * (A.t() + m_cap[:, None] + m_img[None, :]), gate)   # [W,G] :241-259
    i2t = _cos_rows(Q, P @ Ch).mean()                                      # :262-269, 317-318
    # text -> image: normalise each word over regions, softmax over words for each region
    A2 = S / (S.pow(2).sum(0, keepdim=True).sqrt() + 1e-8)
    P2 = _focal_softmax(lam * (A2 + m_img[:, None] + m_cap[None, :]), gate)      # [G,W]
    t2i = _cos_rows(C, P2 @ Qh).mean()                                     # :320-327
    return i2t + t2i


def xattn_scores(images, captions, img_mask, cap_mask, lam=20.0, gate=True):
    """[n_img, n_cap] matrix of pair scores (row = video)."""
    rows = []
    for i in range(images.shape[0]):
        rows.append(torch.stack([xattn_pair(images[i], captions[j], img_mask[i], cap_mask[j], lam, gate)
                                 for j in range(captions.shape[0])]))
    return torch.stack(rows)


def xattn_scores_batched(images, captions, img_mask, cap_mask, lam=20.0, gate=True):
    """Same numbers as ``xattn_scores`` with all pairs at once ([n_i,n_c,G,W] intermediates, as the reference
    materialises them) -- the form timed as the CPU baseline."""
    Ch, Qh = _unit(images), _unit(captions)
    S = F.leaky_relu(torch.einsum("igd,jwd->ijgw", Ch, Qh), 0.1)
    mi, mc = img_mask[:, None, :, None], cap_mask[None, :, None, :]
    A = S / (S.pow(2).sum(3, keepdim=True).sqrt() + 1e-8)
    P = _focal_softmax((lam * (A + mi + mc)).transpose(2, 3), gate)             # [i,j,W,G]
    wc = torch.einsum("ijwg,igd->ijwd", P, Ch)
    i2t = _cos_rows(captions[None], wc).mean(-1)
    A2 = S / (S.pow(2).sum(2, keepdim=True).sqrt() + 1e-8)
    P2 = _focal_softmax(lam * (A2 + mi + mc), gate)                              # [i,j,G,W]
    wc2 = torch.einsum("ijgw,jwd->ijgd", P2, Qh)
    t2i = _cos_rows(images[:, None], wc2).mean(-1)
    return i2t + t2i


def rwa_loss(scores, lam=20.0):
    """model/loss.py:105-116."""
    n = scores.shape[0]
    eye = torch.eye(n, dtype=scores.dtype, device=scores.device)
    z = scores * lam
    return (torch.softmax(z, 1) * (torch.log_softmax(z, 1) - torch.log(eye + 1e-6))).sum(1).mean()


def global_local_loss(out, text_mask_add, lam=20.0, temperature=0.05, gate=True, batched=None):
    """A11 + the trainer's glue (trainer/trainer_dist.py:156-164).  ``out`` is model_forward's dict,
    ``text_mask_add`` = (attention_mask[:,1:] - 1) * 100.  Returns (loss, global, local, sim, xattn)."""
    sim = sim_matrix(out["global_text_embeddings"], out["global_object_embeddings"])
    g = norm_softmax_loss(sim, temperature)
    if batched is None:
        batched = out["local_object_embeddings"].shape[0] > 4      # the all-pairs form (what the reference materialises) beyond a few pairs
    xs = (xattn_scores_batched if batched else xattn_scores)(out["local_object_embeddings"], out["local_text_embeddings"],
                                                            out["object_mask"].to(torch.float32), text_mask_add.to(torch.float32), lam, gate)
    l = rwa_loss(xs, lam)
    return g + l, g, l, sim, xs


# ==================================================================================================
# A14  HF AdamW (transformers.optimization.AdamW 4.10.0, restated from its documented update; parity unpinned)
# ==================================================================================================
def hf_adamw_step(param, grad, m, v, step, lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0):
    """In-place.  eps is added to sqrt(v) BEFORE the bias-correction scaling; decoupled decay after the update."""
    m.mul_(beta1).add_(grad, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    param.addcdiv_(m, denom, value=-step_size)
    if weight_decay > 0.0:
        param.add_(param, alpha=-lr * weight_decay)


# ==================================================================================================
# whole step (used for golden grads and as the timed CPU baseline)
# ==================================================================================================
def train_step(p, input_ids, att_mask01, obj, mask01, gate=True, drop=None):
    """fwd + loss + bwd on a dict of leaf tensors requiring grad.  Returns (loss, global, local)."""
    out = model_forward(p, input_ids, att_mask01, obj, mask01, drop=drop)
    tmask = (att_mask01[:, 1:].to(torch.float32) - 1.0) * 100.0
    loss, g, l, _, _ = global_local_loss(out, tmask, gate=gate)
    loss.backward()
    return loss.detach(), g.detach(), l.detach()


def params_from_numpy(sd, requires_grad=False):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(requires_grad) for k, v in sd.items()}
