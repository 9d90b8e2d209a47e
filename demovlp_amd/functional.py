"""autograd.Function wrappers: one node per tower layer / loss head, each forward and backward a short, explicit
sequence of C-ABI kernel launches (ops.py).  Layer-level nodes (rather than op-level) let the residual adds, GELU,
bias and gradient accumulations live in GEMM / LayerNorm epilogues instead of separate element-wise launches, and give
DDP-style hooks one firing point per layer for bucketed gradient all-reduce.

Master parameters stay fp32; with ``compute dtype = bfloat16`` the GEMM weights are read from bf16 shadows kept by
``SHADOWS`` (refreshed when the parameter's version counter moves, or written directly by the fused optimizer).
Weight gradients are produced in fp32 straight from the MFMA epilogue.
"""
from __future__ import annotations

import os

import torch

from . import ops


class ShadowCache:
    """bf16 copies of fp32 master weights for the MFMA kernels."""

    def __init__(self):
        self._c = {}

    def get(self, w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        if dtype == torch.float32:
            return w.detach()
        e = self._c.get(id(w))
        if e is None or e[0] != w._version or e[1].dtype != dtype or e[2] != w.data_ptr():
            dst = e[1] if (e is not None and e[1].dtype == dtype and e[1].shape == w.shape and not e[3]) else None
            sh = ops.cast(w.detach(), dtype, out=dst)
            e = (w._version, sh, w.data_ptr(), False)
            self._c[id(w)] = e
        return e[1]

    def adopt(self, w: torch.Tensor, shadow_view: torch.Tensor):
        """Register an externally maintained shadow (a view of the fused optimizer's flat bf16 buffer)."""
        self._c[id(w)] = (w._version, shadow_view, w.data_ptr(), True)

    def invalidate(self):
        """The master weights changed behind autograd's back (the fused optimizer writes through raw pointers, so no version
        counter moves): every shadow this cache maintains itself is re-cast on its next use.  Adopted (optimizer-written)
        shadows are current by construction."""
        for k, e in list(self._c.items()):
            if not e[3]:
                self._c[k] = (-1, e[1], e[2], False)

    def clear(self):
        self._c.clear()


SHADOWS = ShadowCache()


def _grad_buf(param):
    """fp32 gradient destination for a parameter: its slice of the flat gradient arena when one is attached."""
    gv = getattr(param, "_dvlp_grad_view", None)
    # a fresh view object each time: autograd's AccumulateGrad adopts (rather than clones) a gradient it solely owns
    return gv.view(gv.shape) if gv is not None else None


OVERLAP_WGRAD = 0         # side-stream gradient work (needs gradient arenas): 0 off, 1 bias-gradient reductions only (they slip in
                          # beside the main stream's GEMMs), 2 also the weight-gradient GEMMs (GEMM || GEMM: higher step throughput,
                          # but each GEMM's own duration stretches, which blurs per-kernel timing)


OVERLAP_TEXT_ONLY = False  # experiment: side-stream gradient work only for launches made on the text tower's stream (its 75-tile products
                           # leave 70 % of the CUs to whatever runs beside them)


class _Side:
    """Context: enqueue on the side stream everything that only feeds parameter gradients.  The tensors it reads were
    produced on the main stream, so the side stream first waits for the main stream's current position, and the caching
    allocator is told they are in use there."""

    def __init__(self, *tensors, level=1):
        self.tensors = tensors
        self.on = OVERLAP_WGRAD >= level and torch.cuda.is_available()
        if self.on and OVERLAP_TEXT_ONLY:
            dev = torch.cuda.current_device()
            self.on = dev in ops._TEXT and ops._TEXT[dev] == torch.cuda.current_stream()

    def __enter__(self):
        if not self.on:
            return self
        main = torch.cuda.current_stream()
        self.side = ops.side_stream()
        self.side.wait_stream(main)
        for t in self.tensors:
            if t is not None:
                t.record_stream(self.side)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False


def join_side_stream():
    """Make the current stream wait for all side-stream work: gradient kernels on the side stream, and the text tower's
    stream when the towers run concurrently (call before gradients are consumed)."""
    if not torch.cuda.is_available():
        return
    cur = torch.cuda.current_stream()
    capturing = torch.cuda.is_current_stream_capturing()

    def joinable(s):
        # While a hipGraph is being captured only streams that were forked INTO the capture may be waited for: an event recorded on a
        # stream outside it (the text tower's stream exists process-wide once any model has used it; this step may not) is an external
        # dependency the captured graph cannot carry -- replays of such a graph came back with corrupted gradients now and then.
        if not capturing:
            return True
        with torch.cuda.stream(s):
            return torch.cuda.is_current_stream_capturing()

    if OVERLAP_WGRAD and joinable(ops.side_stream()):
        cur.wait_stream(ops.side_stream())
    dev = torch.cuda.current_device()
    if dev in ops._TEXT and ops._TEXT[dev] != cur and joinable(ops._TEXT[dev]):
        cur.wait_stream(ops._TEXT[dev])


def _wgrad(dy2d, x2d, param):
    with _Side(dy2d, x2d, level=2):
        return ops.linear_bwd_weight(dy2d, x2d, out=_grad_buf(param))


def _wgrad_group(items):
    """Weight gradients of one layer [(dy2d, x2d, param | fp32 destination tensor), ...] as one grouped launch (ops.wgrad_grouped).  An
    optimizer update left pending by the layer whose backward ran just before on this stream (``_early_update``) rides on the launch's
    spare workgroups."""
    with _Side(*[t for it in items for t in it[:2]], level=2):
        ride = TAKE_PENDING_OPT() if TAKE_PENDING_OPT is not None else None
        return ops.wgrad_grouped([(dy2d, x2d, _grad_buf(dst) if isinstance(dst, torch.nn.Parameter) else dst) for dy2d, x2d, dst in items], ride=ride)


EARLY_OPT = None          # trainer.FusedAdamW.begin_overlapped installs its per-layer update here for the duration of one backward
TAKE_PENDING_OPT = None   # ... and the hand-over of the update it left pending (to ride in the next grouped weight-gradient launch of the same stream)


def _early_update(params, now=False):
    """End of a layer's backward: its weight gradients are final (side stream) and its weights have been read for the last time this
    step (current stream), so the optimizer may update them now -- on the side stream, behind both (``_Side`` waits for the current
    stream).  The HBM-bound update then runs beside the MFMA-bound rest of the backward instead of after it.  ``now``: launch it here
    (no later weight-gradient group on this stream for it to ride in: the embedding tables at the end of the text tower's backward)."""
    if EARLY_OPT is None:
        return
    with _Side(level=2):
        EARLY_OPT(params, now)


FUSE_TEXT_QKV = os.environ.get("DVLP_NO_FUSED_QKV") is None      # developer switch for A/B timing
FUSE_LN_COLSUM = os.environ.get("DVLP_NO_LN_COLSUM") is None     # developer switch for A/B timing
FUSE_GEMM_COLSUM = os.environ.get("DVLP_NO_GEMM_COLSUM") is None  # developer switch for A/B timing
FUSE_ATTN_COLSUM = os.environ.get("DVLP_NO_ATTN_COLSUM") is None  # developer switch for A/B timing


def _stacked(ts):
    """If equally shaped tensors sit back to back in one allocation (parameter-arena order), one [n*rows, ...] view of them."""
    a = ts[0]
    if any(t is None or t.shape != a.shape or t.dtype != a.dtype or not t.is_contiguous() for t in ts):
        return None
    step = a.numel() * a.element_size()
    same = all(t.untyped_storage().data_ptr() == a.untyped_storage().data_ptr() and t.data_ptr() == a.data_ptr() + i * step for i, t in enumerate(ts))
    if not same:
        return None
    shape = (len(ts) * a.shape[0],) + tuple(a.shape[1:])
    stride = tuple(a.stride())
    return torch.as_strided(a, shape, stride)


def _fused_qkv(qw, qb, kw, kb, vw, vb, cd):
    """DistilBERT keeps q_lin / k_lin / v_lin as separate modules; inside a ParamArena their weights (and biases, gradients,
    bf16 shadows) are adjacent, so the three projections run as ONE [2304, 768] linear.  None when they are not adjacent."""
    if not FUSE_TEXT_QKV:
        return None
    W = _stacked([SHADOWS.get(w, cd) for w in (qw, kw, vw)])
    b = _stacked([t.detach() for t in (qb, kb, vb)])
    gW = _stacked([getattr(w, "_dvlp_grad_view", None) for w in (qw, kw, vw)])
    gb = _stacked([getattr(t, "_dvlp_grad_view", None) for t in (qb, kb, vb)])
    if W is None or b is None or gW is None or gb is None:
        return None
    return W, b, gW, gb


def _bgrad(dy2d, param):
    gb = _grad_buf(param)
    with _Side(dy2d):
        # written into the gradient arena: nobody reads it before the trainer's flush, so the final reduction may be batched
        return ops.colsum(dy2d, out=gb, defer=gb is not None)


def _dx_with_bias_grad(dy2d, w_next, gelu_pre, bias):
    """d(pre-activation) = (dy W_next) * gelu'(pre) together with the gradient of the bias that was added to that pre-activation
    (its column sums): out of one GEMM when the bias gradient lives in an arena, else GEMM + column-sum pass."""
    gb = _grad_buf(bias)
    if gb is not None and FUSE_GEMM_COLSUM:
        return ops.linear_bwd_input(dy2d, w_next, gelu_pre=gelu_pre, colsum_to=gb), gb
    dpre = ops.linear_bwd_input(dy2d, w_next, gelu_pre=gelu_pre)
    return dpre, _bgrad(dpre, bias)


def _ln_bwd(dy2d, x2d, w, b, mean, rstd, dres=None, bias_of_next=None):
    """LayerNorm backward with dgamma / dbeta going to the parameters' arena slices (deferred final reduction) when attached.
    ``bias_of_next``: bias parameter of the Linear that consumed the LayerNorm's INPUT... i.e. whose output gradient is the dx
    computed here: its gradient (column sums of dx) then comes out of the same kernel.  Returns (dx, dgamma, dbeta, dbias|None)."""
    gw, gb = _grad_buf(w), _grad_buf(b)
    gnext = _grad_buf(bias_of_next) if bias_of_next is not None else None
    fused = gnext is not None and gw is not None and gb is not None and FUSE_LN_COLSUM
    dx, dg, db = ops.layernorm_bwd(dy2d, x2d, w.detach(), mean, rstd, dres=dres, out_gamma=gw, out_beta=gb,
                                   defer=gw is not None and gb is not None, dx_colsum=gnext if fused else None)
    if bias_of_next is None:
        return dx, dg, db
    return dx, dg, db, (gnext if fused else _bgrad(dx, bias_of_next))


def _into(param, value):
    """Place a small fp32 gradient into the parameter's arena slice (if any) and return the tensor to hand to autograd."""
    gv = _grad_buf(param)
    if gv is None:
        return value.reshape(param.shape)
    ops.copy_by_kernel(gv, value.reshape(param.shape))       # (no memcpy nodes in a captured step: see ops.copy_by_kernel)
    return gv


# ----------------------------------------------------------------------------------------------------------------
# the heads' view of a tower output: (x[:, 0], x[:, 1:]), contiguous (model/model.py:70-96)
# ----------------------------------------------------------------------------------------------------------------
class SplitClsFn(torch.autograd.Function):
    """x [B,N,d] -> (global [B,d], local [B,N-1,d]) in one launch; backward assembles dx from the two gradients in one launch
    (autograd's own backward of ``x[:, 0].contiguous(), x[:, 1:].contiguous()``: two zero fills, two strided copies, an add)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.meta = (tuple(x.shape), x.dtype, x.device)
        return ops.split_cls(x)

    @staticmethod
    def backward(ctx, dg, dl):
        (B, N, d), dtype, device = ctx.meta
        dg = None if dg is None else dg.to(dtype).contiguous()
        dl = None if dl is None else dl.to(dtype).contiguous()
        return ops.merge_cls(dg, dl, B, N, d, dtype, device)


# ----------------------------------------------------------------------------------------------------------------
# generic linear (object_model.proj, txt_proj)
# ----------------------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = x W^T + b.  With ``relu_src`` given, x must be relu(relu_src) and the gradient is returned for relu_src."""

    @staticmethod
    def forward(ctx, x, w, b, relu_src):
        cd = x.dtype
        wc = SHADOWS.get(w, cd)
        x2 = x.reshape(-1, x.shape[-1])
        y = ops.linear_fwd(x2, wc, b.detach() if b is not None else None)
        ctx.save_for_backward(x2, w, relu_src.reshape(x2.shape) if relu_src is not None else None)
        ctx.has_b = b is not None
        ctx.bparam = b
        ctx.in_shape = x.shape
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, relu_src = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        wc = SHADOWS.get(w, x2.dtype)
        dx = ops.linear_bwd_input(dy2, wc, relu_pre=relu_src)
        dw = _wgrad(dy2, x2, w)
        db = _bgrad(dy2, ctx.bparam) if ctx.has_b else None
        dx = dx.reshape(ctx.in_shape)
        if relu_src is not None:
            return None, dw, db, dx
        return dx, dw, db, None


# ----------------------------------------------------------------------------------------------------------------
# object tower
# ----------------------------------------------------------------------------------------------------------------
class ObjectPrologueFn(torch.autograd.Function):
    """model/object_transformer.py:400-433: region/box embedding, CLS, temporal embedding, additive mask."""

    @staticmethod
    def forward(ctx, obj, mask01, Wo, bo, Wp, bp, temporal, cls, pos_embed, cd):
        B, F, R, _ = obj.shape
        feat, box = ops.obj_split(obj, cd)
        tok = ops.linear_fwd(feat, SHADOWS.get(Wo, cd), bo.detach())
        pos0 = pos_embed.detach()[0, 0].contiguous()
        # the FULL temporal_embed parameter comes in (curr_frames < num_frames uses its first F rows, object_transformer.py:423-432),
        # so its gradient lands in the parameter's own arena slice like every other one
        x, addmask = ops.embed_assemble(tok, box, Wp.detach(), bp.detach(), temporal.detach()[0, :F].contiguous(), cls.detach().reshape(768),
                                        pos0, mask01, B, F, R)
        ctx.save_for_backward(feat, box)
        ctx.params = (Wo, bo, Wp, bp, temporal, cls, pos_embed)
        ctx.dims = (B, F, R)
        ctx.mark_non_differentiable(addmask)
        return x.reshape(B, 1 + F * R, 768), addmask

    @staticmethod
    def backward(ctx, dx, _dmask):
        feat, box = ctx.saved_tensors
        Wo, bo, Wp, bp, temporal, cls, pos_embed = ctx.params
        B, F, R = ctx.dims
        N = 1 + F * R
        dx = dx.contiguous()
        dtok = ops.embed_unassemble(dx, B, F, R)
        dWo = _wgrad(dtok, feat, Wo)
        dbo = _bgrad(dtok, bo)
        dbp = _bgrad(dtok, bp)                            # same sum (both biases are added to every region token), but dbo may
                                                          # still be a deferred reduction here: it cannot be read and copied
        dWp = ops.box_wgrad(dtok, box, out=_grad_buf(Wp))
        # temporal[f] = sum over (b, r) of dtok[b, f, r, :]
        gt = _grad_buf(temporal)
        if gt is None:
            gt = torch.empty_like(temporal, dtype=torch.float32)
        if temporal.shape[1] > F:
            gt[0, F:].zero_()                             # frames beyond the clip length receive no gradient
        ops.colsum_grouped(dtok, B * R, 768, 768, R, F * R * 768, F, R * 768, out=gt[0, :F])
        dtemp = gt
        dcls_row = ops.colsum_grouped(dx, B, 768, N * 768, B, 0, 1, 0).reshape(768)     # sum_b dx[b, 0, :]
        dcls = _into(cls, dcls_row.reshape(1, 1, 768))
        dpos = torch.zeros_like(pos_embed)
        ops.copy_by_kernel(dpos[0, 0], dcls_row)          # (dpos[0, 0] = dcls_row would be a memcpy node in a captured step)
        dpos = _into(pos_embed, dpos)
        return None, None, dWo, dbo, dWp, dbp, dtemp, dcls, dpos, None


class VitBlockFn(torch.autograd.Function):
    """SpaceTimeBlock with time_module falsy (model/object_transformer.py:249-274): pre-LN space attention + MLP."""

    @staticmethod
    def forward(ctx, x, addmask, n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b, F, R, f2b_below=None, f2b_from_above=False):
        """``f2b_below`` / ``f2b_from_above`` (ObjectTransformer.forward_features, blocks chained output -> input): the gradient
        entering the block below IS the dx this block's norm1 backward writes, so the column sums of that dx -- the gradient of the
        lower block's fc2 bias -- come out of the LayerNorm-backward kernel that produces it, and the lower block skips its own
        column-sum pass over the same 28 MB (12 launches of ~11 us per step)."""
        B, N, D = x.shape
        cd = x.dtype
        x2 = x.reshape(B * N, D)
        h1, _, m1, r1 = ops.layernorm_fwd(x2, n1w.detach(), n1b.detach(), 1e-6)
        qkv = ops.linear_fwd(h1, SHADOWS.get(qkvw, cd), qkvb.detach())
        att, cls_stats = ops.space_attention_fwd(qkv, addmask, B, F, R, want_stats=True)
        x1 = ops.linear_fwd(att, SHADOWS.get(pw, cd), pb.detach(), res=x2)
        h2, _, m2, r2 = ops.layernorm_fwd(x1, n2w.detach(), n2b.detach(), 1e-6)
        pre = torch.empty((B * N, f1w.shape[0]), device=x.device, dtype=cd)
        a = ops.linear_fwd(h2, SHADOWS.get(f1w, cd), f1b.detach(), gelu_aux=pre)
        y = ops.linear_fwd(a, SHADOWS.get(f2w, cd), f2b.detach(), res=x1)
        ctx.save_for_backward(x2, addmask, m1, r1, h1, qkv, att, x1, m2, r2, h2, pre, a, cls_stats)
        ctx.params = (n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b)
        ctx.dims = (B, N, F, R)
        ctx.f2b_below, ctx.f2b_from_above = f2b_below, bool(f2b_from_above)
        return y.reshape(B, N, D)

    @staticmethod
    def backward(ctx, dy):
        x2, addmask, m1, r1, h1, qkv, att, x1, m2, r2, h2, pre, a, cls_stats = ctx.saved_tensors
        n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b = ctx.params
        B, N, F, R = ctx.dims
        cd = x2.dtype
        dy2 = dy.reshape(B * N, -1).contiguous()
        # the block above already queued colsum(dy) for this bias in its norm1 backward (same tensor: its dx is this dy)
        df2b = _grad_buf(f2b) if ctx.f2b_from_above else _bgrad(dy2, f2b)
        dpre, df1b = _dx_with_bias_grad(dy2, SHADOWS.get(f2w, cd), pre, f1b)
        dh2 = ops.linear_bwd_input(dpre, SHADOWS.get(f1w, cd))
        dx1, dn2w, dn2b, dpb = _ln_bwd(dh2, x1, n2w, n2b, m2, r2, dres=dy2, bias_of_next=pb)
        datt = ops.linear_bwd_input(dx1, SHADOWS.get(pw, cd))
        gqb = _grad_buf(qkvb)
        if gqb is not None and FUSE_ATTN_COLSUM:       # qkv-bias gradient = column sums of dqkv: queued from inside the attention backward
            dqkv, fused = ops.space_attention_bwd(qkv, addmask, datt, B, F, R, out=att, stats=cls_stats, colsum_to=gqb)
            dqkvb = gqb if fused else _bgrad(dqkv, qkvb)
        else:
            dqkv = ops.space_attention_bwd(qkv, addmask, datt, B, F, R, out=att, stats=cls_stats)
            dqkvb = _bgrad(dqkv, qkvb)
        dh1 = ops.linear_bwd_input(dqkv, SHADOWS.get(qkvw, cd))
        # the four weight gradients of the block (9-36 output tiles each, K = B*N tokens) as ONE grouped GEMM
        df2w, df1w, dpw, dqkvw = _wgrad_group([(dy2, a, f2w), (dpre, h2, f1w), (dx1, att, pw), (dqkv, h1, qkvw)])
        if ctx.f2b_below is not None:
            dx, dn1w, dn1b, _ = _ln_bwd(dh1, x2, n1w, n1b, m1, r1, dres=dx1, bias_of_next=ctx.f2b_below)
        else:
            dx, dn1w, dn1b = _ln_bwd(dh1, x2, n1w, n1b, m1, r1, dres=dx1)
        _early_update((qkvw, pw, f1w, f2w))
        return (dx.reshape(B, N, -1), None, dn1w, dn1b, dqkvw, dqkvb, dpw, dpb, dn2w, dn2b, df1w, df1b, df2w, df2b, None, None, None, None)


class TimeSpaceBlockFn(torch.autograd.Function):
    """SpaceTimeBlock with time_module='timeattn' (model/object_transformer.py:249-274): time attention over norm3(x) for every
    region slot ('b (f n) d -> (b n) f d', CLS query over all tokens), space attention over norm1(x + time), and -- the
    FrozenInTime residual -- the space output is added to the block INPUT x (:267), then the MLP.  The time attention is the
    space-attention kernel on the token grid transposed to region-major order (ops.token_transpose) with (F, R) swapped."""

    @staticmethod
    def forward(ctx, x, addmask, addmask_t, n3w, n3b, tqw, tqb, tpw, tpb, n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b, F, R):
        B, N, D = x.shape
        cd = x.dtype
        x2 = x.reshape(B * N, D)
        h3, _, m3, r3 = ops.layernorm_fwd(x2, n3w.detach(), n3b.detach(), 1e-6)
        h3t = ops.token_transpose(h3.reshape(B, N, D), B, F, R).reshape(B * N, D)
        tqkv = ops.linear_fwd(h3t, SHADOWS.get(tqw, cd), tqb.detach())
        tatt = ops.space_attention_fwd(tqkv, addmask_t, B, R, F)                   # frames <-> regions: time attention
        tout = ops.linear_fwd(tatt, SHADOWS.get(tpw, cd), tpb.detach())            # region-major rows
        xin = ops.token_transpose(tout.reshape(B, N, D), B, R, F, res=x).reshape(B * N, D)      # back to frame-major, + x
        h1, _, m1, r1 = ops.layernorm_fwd(xin, n1w.detach(), n1b.detach(), 1e-6)
        qkv = ops.linear_fwd(h1, SHADOWS.get(qkvw, cd), qkvb.detach())
        att = ops.space_attention_fwd(qkv, addmask, B, F, R)
        x1 = ops.linear_fwd(att, SHADOWS.get(pw, cd), pb.detach(), res=x2)         # residual from x, not from xin
        h2, _, m2, r2 = ops.layernorm_fwd(x1, n2w.detach(), n2b.detach(), 1e-6)
        pre = torch.empty((B * N, f1w.shape[0]), device=x.device, dtype=cd)
        a = ops.linear_fwd(h2, SHADOWS.get(f1w, cd), f1b.detach(), gelu_aux=pre)
        y = ops.linear_fwd(a, SHADOWS.get(f2w, cd), f2b.detach(), res=x1)
        ctx.save_for_backward(x2, addmask, addmask_t, m3, r3, h3t, tqkv, tatt, xin, m1, r1, h1, qkv, att, x1, m2, r2, h2, pre, a)
        ctx.params = (n3w, n3b, tqw, tqb, tpw, tpb, n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b)
        ctx.dims = (B, N, F, R)
        return y.reshape(B, N, D)

    @staticmethod
    def backward(ctx, dy):
        x2, addmask, addmask_t, m3, r3, h3t, tqkv, tatt, xin, m1, r1, h1, qkv, att, x1, m2, r2, h2, pre, a = ctx.saved_tensors
        n3w, n3b, tqw, tqb, tpw, tpb, n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b = ctx.params
        B, N, F, R = ctx.dims
        cd = x2.dtype
        D = x2.shape[1]
        dy2 = dy.reshape(B * N, -1).contiguous()
        df2b = _bgrad(dy2, f2b)
        dpre, df1b = _dx_with_bias_grad(dy2, SHADOWS.get(f2w, cd), pre, f1b)
        dh2 = ops.linear_bwd_input(dpre, SHADOWS.get(f1w, cd))
        dx1, dn2w, dn2b, dpb = _ln_bwd(dh2, x1, n2w, n2b, m2, r2, dres=dy2, bias_of_next=pb)
        datt = ops.linear_bwd_input(dx1, SHADOWS.get(pw, cd))
        dqkv = ops.space_attention_bwd(qkv, addmask, datt, B, F, R)
        dqkvb = _bgrad(dqkv, qkvb)
        dh1 = ops.linear_bwd_input(dqkv, SHADOWS.get(qkvw, cd))
        dxin, dn1w, dn1b = _ln_bwd(dh1, xin, n1w, n1b, m1, r1)                      # xin feeds norm1 only
        dtout = ops.token_transpose(dxin.reshape(B, N, D), B, F, R).reshape(B * N, D)           # gradient of the region-major rows
        dtpb = _bgrad(dtout, tpb)
        dtatt = ops.linear_bwd_input(dtout, SHADOWS.get(tpw, cd))
        dtqkv = ops.space_attention_bwd(tqkv, addmask_t, dtatt, B, R, F)
        dtqb = _bgrad(dtqkv, tqb)
        dh3t = ops.linear_bwd_input(dtqkv, SHADOWS.get(tqw, cd))
        dh3 = ops.token_transpose(dh3t.reshape(B, N, D), B, R, F).reshape(B * N, D)
        dres = ops.token_transpose(dxin.reshape(B, N, D), B, 1, N - 1, res=dx1.reshape(B, N, D)).reshape(B * N, D)   # dx1 + dxin (x feeds both residuals)
        df2w, df1w, dpw, dqkvw, dtpw, dtqw = _wgrad_group([(dy2, a, f2w), (dpre, h2, f1w), (dx1, att, pw), (dqkv, h1, qkvw),
                                                           (dtout, tatt, tpw), (dtqkv, h3t, tqw)])
        dx, dn3w, dn3b = _ln_bwd(dh3, x2, n3w, n3b, m3, r3, dres=dres)
        _early_update((tqw, tpw, qkvw, pw, f1w, f2w))
        return (dx.reshape(B, N, -1), None, None, dn3w, dn3b, dtqw, dtqb, dtpw, dtpb, dn1w, dn1b, dqkvw, dqkvb, dpw, dpb, dn2w, dn2b,
                df1w, df1b, df2w, df2b, None, None)


# ----------------------------------------------------------------------------------------------------------------
# text tower (DistilBERT)
# ----------------------------------------------------------------------------------------------------------------
class TextEmbedFn(torch.autograd.Function):
    """word + position embeddings -> LayerNorm(eps 1e-12).  padding_idx 0 receives no gradient."""

    @staticmethod
    def forward(ctx, ids, word, pos, lnw, lnb, cd, drop=None):
        """``drop``: None or (p, state, site): HF Embeddings' dropout after the LayerNorm (train mode)."""
        e, y, mean, rstd = ops.text_embed_fwd(ids, word.detach(), pos.detach(), lnw.detach(), lnb.detach(), 1e-12, cd)
        keep = None
        if drop is not None:
            y, keep = ops.dropout_fwd(y, drop[0], drop[1], drop[2])
        ctx.save_for_backward(ids, e, mean, rstd, keep)
        ctx.params = (word, pos, lnw, lnb)
        ctx.drop_p = drop[0] if drop is not None else 0.0
        return y.reshape(ids.shape[0], ids.shape[1], 768)

    @staticmethod
    def backward(ctx, dy):
        ids, e, mean, rstd, keep = ctx.saved_tensors
        word, pos, lnw, lnb = ctx.params
        B, L = ids.shape
        dy2 = dy.reshape(B * L, 768).contiguous()
        if keep is not None:
            dy2 = ops.dropout_bwd(dy2, keep, ctx.drop_p)
        de, dg, db = _ln_bwd(dy2, e, lnw, lnb, mean, rstd)
        gv = _grad_buf(word)
        if gv is not None:
            gv.zero_()
            ops.call("dvlp_text_embed_bwd", ops.dt(de), B * L, ops.p(ids), ops.p(de), ops.p(gv), ops.stream())
            dword = gv
        else:
            dword = ops.text_embed_bwd(ids, de, word.shape[0])
        dpos = torch.zeros_like(pos)
        ops.colsum_grouped(de, B, 768, L * 768, B, 0, L, 768, out=dpos[:L])       # sum over the batch per position, written in place
        dpos = _into(pos, dpos)
        # (round 6, measured and not kept: updating the two embedding tables here -- 0.7 GB of optimizer traffic on the text tower's stream
        #  beside the object tower's backward instead of in the step's tail -- left the replayed step where it was: 16.99 / 16.97 / 17.00
        #  against 16.97 / 16.99 / 17.01 ms)
        return None, dword, dpos, dg, db, None, None


class BertLayerFn(torch.autograd.Function):
    """One DistilBERT TransformerBlock (post-LN, eps 1e-12, exact GELU).  ``want_relu``: also return relu(y)."""

    @staticmethod
    def forward(ctx, x, addmask, qw, qb, kw, kb, vw, vb, ow, ob, l1w, l1b, f1w, f1b, f2w, f2b, l2w, l2b, want_relu, drop=None):
        """``drop``: None or (p_attention, p_hidden, state, site): HF's train-mode dropouts of this block -- on the attention
        probabilities (site) and on the feed-forward output before its residual (site + 1)."""
        B, L, D = x.shape
        cd = x.dtype
        x2 = x.reshape(B * L, D)
        akeep = None
        if drop is not None and drop[0] > 0.0:
            akeep = ops.attn_keep_masks(B, L, drop[0], drop[2], drop[3], x.device)
        fused = _fused_qkv(qw, qb, kw, kb, vw, vb, cd)
        if fused is not None:
            qkv = ops.linear_fwd(x2, fused[0], fused[1])                          # [B*L, 2304] = q | k | v
            q, k, v = qkv[:, :768], qkv[:, 768:1536], qkv[:, 1536:]
            att = ops.full_attention_fwd(q, k, v, addmask, B, L, ld=2304, keep=akeep)
        else:
            q = ops.linear_fwd(x2, SHADOWS.get(qw, cd), qb.detach())
            k = ops.linear_fwd(x2, SHADOWS.get(kw, cd), kb.detach())
            v = ops.linear_fwd(x2, SHADOWS.get(vw, cd), vb.detach())
            att = ops.full_attention_fwd(q, k, v, addmask, B, L, keep=akeep)
        ctx.fused = fused is not None
        s1 = ops.linear_fwd(att, SHADOWS.get(ow, cd), ob.detach(), res=x2)
        x1, _, m1, r1 = ops.layernorm_fwd(s1, l1w.detach(), l1b.detach(), 1e-12)
        pre = torch.empty((B * L, f1w.shape[0]), device=x.device, dtype=cd)
        a = ops.linear_fwd(x1, SHADOWS.get(f1w, cd), f1b.detach(), gelu_aux=pre)
        fkeep = None
        if drop is not None and drop[1] > 0.0:
            f = ops.linear_fwd(a, SHADOWS.get(f2w, cd), f2b.detach())
            s2, fkeep = ops.dropout_fwd(f, drop[1], drop[2], drop[3] + 1, res=x1)     # dropout(lin2(.)) + residual
        else:
            s2 = ops.linear_fwd(a, SHADOWS.get(f2w, cd), f2b.detach(), res=x1)
        y, yr, m2, r2 = ops.layernorm_fwd(s2, l2w.detach(), l2b.detach(), 1e-12, want_relu=want_relu)
        ctx.save_for_backward(x2, addmask, q, k, v, att, s1, m1, r1, x1, pre, a, s2, m2, r2, fkeep,
                              *(akeep[:2] if akeep is not None else (None, None)))
        ctx.drop = (drop[0], drop[1]) if drop is not None else (0.0, 0.0)
        ctx.params = (qw, qb, kw, kb, vw, vb, ow, ob, l1w, l1b, f1w, f1b, f2w, f2b, l2w, l2b)
        ctx.dims = (B, L)
        y = y.reshape(B, L, D)
        if want_relu:
            yr = yr.reshape(B, L, D)
            ctx.mark_non_differentiable(yr)
            return y, yr
        return y, None

    @staticmethod
    def backward(ctx, dy, _dyr):
        x2, addmask, q, k, v, att, s1, m1, r1, x1, pre, a, s2, m2, r2, fkeep, akeep, akeepT = ctx.saved_tensors
        qw, qb, kw, kb, vw, vb, ow, ob, l1w, l1b, f1w, f1b, f2w, f2b, l2w, l2b = ctx.params
        B, L = ctx.dims
        cd = x2.dtype
        dy2 = dy.reshape(B * L, -1).contiguous()
        keep = (akeep, akeepT, 1.0 / (1.0 - ctx.drop[0])) if akeep is not None else None
        if fkeep is not None:
            ds2, dl2w, dl2b = _ln_bwd(dy2, s2, l2w, l2b, m2, r2)
            dff = ops.dropout_bwd(ds2, fkeep, ctx.drop[1])          # gradient of lin2's output: through the dropout mask
            df2b = _bgrad(dff, f2b)
        else:
            ds2, dl2w, dl2b, df2b = _ln_bwd(dy2, s2, l2w, l2b, m2, r2, bias_of_next=f2b)
            dff = ds2
        dpre, df1b = _dx_with_bias_grad(dff, SHADOWS.get(f2w, cd), pre, f1b)
        dx1 = ops.linear_bwd_input(dpre, SHADOWS.get(f1w, cd), res=ds2)
        ds1, dl1w, dl1b, dob = _ln_bwd(dx1, s1, l1w, l1b, m1, r1, bias_of_next=ob)
        datt = ops.linear_bwd_input(ds1, SHADOWS.get(ow, cd))
        fused = _fused_qkv(qw, qb, kw, kb, vw, vb, cd) if ctx.fused else None
        if fused is not None:
            W, _, gW, gb = fused
            dqkv = torch.empty((B * L, 2304), device=q.device, dtype=cd)
            ops.full_attention_bwd(q, k, v, addmask, datt, B, L, ld=2304, out=(dqkv[:, :768], dqkv[:, 768:1536], dqkv[:, 1536:]), ld_out=2304, keep=keep)
            with _Side(dqkv):
                ops.colsum(dqkv, out=gb, defer=True)
            df2w, df1w, dow, _ = _wgrad_group([(dff, a, f2w), (dpre, x1, f1w), (ds1, att, ow), (dqkv, x2, gW)])
            dqw, dkw, dvw, dqb, dkb, dvb = (_grad_buf(t) for t in (qw, kw, vw, qb, kb, vb))      # slices of the fused gradients
            dx = ops.linear_bwd_input(dqkv, W, res=ds1)
        else:
            dq, dk, dv = ops.full_attention_bwd(q, k, v, addmask, datt, B, L, keep=keep)
            dqb, dkb, dvb = _bgrad(dq, qb), _bgrad(dk, kb), _bgrad(dv, vb)
            df2w, df1w, dow, dqw, dkw, dvw = _wgrad_group([(dff, a, f2w), (dpre, x1, f1w), (ds1, att, ow), (dq, x2, qw), (dk, x2, kw), (dv, x2, vw)])
            dx = ops.linear_bwd_input(dq, SHADOWS.get(qw, cd), res=ds1)
            ops.linear_bwd_input(dk, SHADOWS.get(kw, cd), out=dx, accumulate=True)
            ops.linear_bwd_input(dv, SHADOWS.get(vw, cd), out=dx, accumulate=True)
        _early_update((qw, kw, vw, ow, f1w, f2w))
        return (dx.reshape(B, L, -1), None, dqw, dqb, dkw, dkb, dvw, dvb, dow, dob, dl1w, dl1b, df1w, df1b, df2w, df2b, dl2w, dl2b, None, None)


# ----------------------------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------------------------
class SimMatrixFn(torch.autograd.Function):
    """model/model.py:582-590 on [N,256] x [M,256].  The per-rank training shape (N == M <= 64) runs inside the one-workgroup
    loss-head kernel; anything else (gathered negatives, the whole eval set) normalises rows once and uses the exact-fp32 GEMM."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        if a.dim() != 2 or b.dim() != 2 or a.shape[1] != 256 or b.shape[1] != 256:
            raise ops._lib.DemoVLPHipError(f"sim_matrix expects [N,256] x [M,256], got {tuple(a.shape)} x {tuple(b.shape)}")
        if a.dtype != b.dtype:
            b = b.to(a.dtype)
        ops.require_gpu(a, b)
        N, M = a.shape[0], b.shape[0]
        ctx.fused = N == M and N <= 64
        if ctx.fused:
            r = ops.global_local_loss(a, b, None, 0.05, 20.0, 1, 0, 1)
            ctx.save_for_backward(a, b)
            return r["sim"]
        an, na = ops.rownorm_fwd(a)
        bn, nb = ops.rownorm_fwd(b)
        ctx.save_for_backward(a, b, an, bn, na, nb)
        return ops.gemm(an, bn, N, M, 256, dtype=ops.F32)

    @staticmethod
    def backward(ctx, dsim):
        dsim = dsim.contiguous().float()
        if ctx.fused:
            a, b = ctx.saved_tensors
            r = ops.global_local_loss(a, b, None, 0.05, 20.0, 1, 0, 4, dsim=dsim)
            return r["dgt"], r["dgo"]
        a, b, an, bn, na, nb = ctx.saved_tensors
        N, M = a.shape[0], b.shape[0]
        dan = ops.gemm(dsim, bn, N, 256, M, trans_b=True, ldb=256, dtype=ops.F32)                  # dsim   [N,M] . bn [M,256]
        dbn = ops.gemm(dsim, an, M, 256, N, trans_a=True, lda=M, trans_b=True, ldb=256, dtype=ops.F32)   # dsim^T [M,N] . an [N,256]
        return ops.rownorm_bwd(a, na, dan), ops.rownorm_bwd(b, nb, dbn)


class NormSoftmaxFn(torch.autograd.Function):
    """model/loss.py:126-138 on a given similarity matrix."""

    @staticmethod
    def forward(ctx, sim, temperature):
        r = ops.global_local_loss(None, None, None, temperature, 20.0, 1, 0, 2, sim=sim.contiguous().float())
        ctx.save_for_backward(r["dsim"])
        return r["losses"][1]

    @staticmethod
    def backward(ctx, g):
        (dsim,) = ctx.saved_tensors
        return dsim * g, None


class XattnFn(torch.autograd.Function):
    """xattn_score_fast (model/loss.py:294-330): [n_img, n_cap] score matrix."""

    @staticmethod
    def forward(ctx, im, s, im_m, s_m, lam, gate):
        im, s = im.contiguous(), s.contiguous()
        im_m = im_m.contiguous().float()
        s_m = s_m.contiguous().float()
        need = im.requires_grad or s.requires_grad
        scores, ws = ops.xattn_fwd(im, s, im_m, s_m, lam, gate, need)
        if need:
            ctx.save_for_backward(im, s, im_m, s_m)
            ctx.ws = ws
            ctx.cfg = (lam, gate)
        return scores

    @staticmethod
    def backward(ctx, dscores):
        im, s, im_m, s_m = ctx.saved_tensors
        lam, gate = ctx.cfg
        dC, dQ = ops.xattn_bwd(im, s, im_m, s_m, lam, gate, dscores.contiguous().float(), ctx.ws)
        ctx.ws = None
        return dC, dQ, None, None, None, None


class RWATailFn(torch.autograd.Function):
    """model/loss.py:105-116 on a given score matrix."""

    @staticmethod
    def forward(ctx, scores, lam):
        r = ops.global_local_loss(None, None, scores.contiguous().float(), 0.05, lam, 0, 1, 2)
        ctx.save_for_backward(r["dxs"])
        return r["losses"][2]

    @staticmethod
    def backward(ctx, g):
        (dxs,) = ctx.saved_tensors
        return dxs * g, None
