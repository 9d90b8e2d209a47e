"""Training-step plumbing around the hot path (mirror of the pieces of trainer/trainer_dist.py the path needs).

* ``AllGather_multi``   -- trainer/trainer_dist.py:13-31, same forward/backward semantics (backward = local slice).
* ``ParamArena``        -- lays every parameter, gradient and AdamW moment out in flat HBM buffers (153 M floats each;
                           trivial next to 288 GB) so the optimizer is ONE fused launch and gradient buckets are
                           contiguous slices that RCCL can reduce in place.
* ``FusedAdamW``        -- transformers.AdamW semantics (train_dist_multi.py:64) on the arena, one HIP kernel.
* ``GradReducer``       -- data-parallel gradient all-reduce over RCCL: contiguous arena buckets launched from
                           post-accumulate hooks while backward is still running (RCCL runs on its own HIP stream),
                           the 26 never-used tensors excluded after the first step instead of find_unused_parameters.
* ``train_step``        -- trainer/trainer_dist.py:138-171 for one batch.
"""
from __future__ import annotations

import warnings

import torch
import torch.distributed as dist

from . import functional as Fn
from . import ops
from .model import sim_matrix


class AllGather_multi(torch.autograd.Function):
    """An autograd function that performs allgather on a tensor (trainer/trainer_dist.py:13-31)."""

    @staticmethod
    def forward(ctx, tensor, n_gpu, args):
        output = [torch.empty_like(tensor) for _ in range(args.world_size)]
        dist.all_gather(output, tensor.contiguous())
        ctx.rank = args.rank
        ctx.batch_size = tensor.shape[0]
        return torch.cat(output, 0)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output[ctx.batch_size * ctx.rank:ctx.batch_size * (ctx.rank + 1)], None, None


class ParamArena:
    ALIGN = 64   # elements; keeps every slice 256-byte aligned in fp32 and 128-byte aligned in bf16

    def __init__(self, module: torch.nn.Module, device=None, bf16_shadow: bool = False):
        # matrices first, vectors (biases, LayerNorm parameters) last: the vectors' gradients are finished by ONE batched
        # reduction at the end of backward (ops.flush_reductions), so they form the tail of the arena -- and of the gradient
        # buckets -- while every bucket of weight gradients is complete, and can be all-reduced, as soon as backward passes it
        named = list(module.named_parameters())
        named = [kv for kv in named if kv[1].dim() >= 2] + [kv for kv in named if kv[1].dim() < 2]
        self.params = [p for _, p in named]
        self.names = [n for n, _ in named]
        pos = {id(p): i for i, p in enumerate(self.params)}
        # arena index of the k-th trainable parameter in module.parameters() order (the order optimizer checkpoints use)
        self.model_order = [pos[id(p)] for p in module.parameters() if p.requires_grad]
        self.n_matrix = sum(1 for p in self.params if p.dim() >= 2)
        self._clean = set()       # slices known to hold zeros (tensors that never receive a gradient are zeroed once, not per step)
        device = device or self.params[0].device
        offs, o = [], 0
        for p in self.params:
            offs.append(o)
            o += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.offsets, self.total = offs, o
        self.vec_lo = offs[self.n_matrix] if self.n_matrix < len(offs) else o
        self.flat_p = torch.zeros(o, device=device, dtype=torch.float32)
        self.flat_g = torch.zeros(o, device=device, dtype=torch.float32)
        self.flat_s = torch.zeros(o, device=device, dtype=torch.bfloat16) if bf16_shadow else None
        for p, off in zip(self.params, offs):
            v = self.flat_p[off:off + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            p._dvlp_grad_view = self.flat_g[off:off + p.numel()].view(p.shape)
            p.grad = None
        if bf16_shadow:
            self.refresh_shadow()

    def refresh_shadow(self):
        ops.cast(self.flat_p, torch.bfloat16, out=self.flat_s)
        self.adopt_shadow()

    def adopt_shadow(self):
        for p, off in zip(self.params, self.offsets):
            Fn.SHADOWS.adopt(p, self.flat_s[off:off + p.numel()].view(p.shape))

    def slice_of(self, i):
        return self.offsets[i], self.offsets[i] + self.params[i].numel()

    def zero_untouched(self, touched):
        """Gradient slices of tensors that received no gradient this step must read as zero for the flat update.  A slice
        zeroed once stays zero until its tensor gets a gradient again, so the ~26 grad-less tensors cost nothing per step."""
        for i in range(len(self.params)):
            if touched(i):
                self._clean.discard(i)
            elif i not in self._clean:
                lo, hi = self.slice_of(i)
                self.flat_g[lo:hi].zero_()
                self._clean.add(i)


class FusedAdamW:
    """HF AdamW (eps added to sqrt(v) before bias correction, decoupled decay, correct_bias=True) over a ParamArena.

    ``state_dict()`` / ``load_state_dict()`` speak the layout the reference checkpoints carry (base/base_trainer.py:185-192
    saves ``optimizer.state_dict()`` of transformers.AdamW built over ``filter(requires_grad, model.parameters())``,
    train_dist_multi.py:60-64): ``{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{..., 'params': [0..n-1]}]}``
    with i counting the trainable parameters in ``model.parameters()`` order; tensors that never received a gradient have no
    entry, as with the lazily created HF state."""

    def __init__(self, arena: ParamArena, lr=1e-5, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if not correct_bias:
            raise NotImplementedError("correct_bias=False is not used by the reference (train_dist_multi.py:64 keeps the default)")
        self.arena = arena
        self.param_groups = [dict(params=arena.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=True)]
        self.m = torch.zeros_like(arena.flat_p)
        self.v = torch.zeros_like(arena.flat_p)
        self.step_count = 0
        self._ever = set()            # arena indices that have received a gradient at least once (own HF-style state)
        self._backwards = {}
        self._hyper, self._hyper_host, self._hyper_step = None, None, 0
        self._early = None            # (lo, hi) ranges already updated in this step (begin_overlapped), else None
        self._index_of = {id(p): i for i, p in enumerate(arena.params)}
        if arena.flat_p.is_cuda:
            ops.enable_deferred_reductions(arena.flat_p.device)
        # gradients are WRITTEN (not accumulated) into the arena by the backward kernels: a second backward() before step()
        # would overwrite the first one's gradients, so count them (two sentinel tensors, one per tower) and refuse
        want = ("object_model.proj.weight", "txt_proj.1.weight")
        for i in ([i for i, n in enumerate(arena.names) if n.endswith(want)] or [0]):
            arena.params[i].register_post_accumulate_grad_hook(lambda _p, i=i: self._backwards.__setitem__(i, self._backwards.get(i, 0) + 1))

    def zero_grad(self, set_to_none=True):
        for p in self.arena.params:
            p.grad = None
        self._backwards = {}
        if self._early is not None:
            raise RuntimeError("FusedAdamW: begin_overlapped() without a matching step() / launch()")

    def _adopt_stray_grads(self):
        """Every gradient must live in its arena slice for the flat update (and the bucketed all-reduce).  The kernels write
        there directly; a gradient autograd materialised elsewhere (a parameter used through a plain torch op) is copied in."""
        for p in self.arena.params:
            g = p.grad
            if g is not None and g.data_ptr() != p._dvlp_grad_view.data_ptr():
                ops.copy_by_kernel(p._dvlp_grad_view, g)

    def prepare(self):
        """Host-side part of a step: everything that inspects autograd state.  After it the flat gradient buffer is final."""
        if any(c > 1 for c in self._backwards.values()):
            raise RuntimeError("FusedAdamW: more than one backward() since zero_grad(): the arena path writes gradients in place and "
                               "does not accumulate across backward passes (scale the loss / enlarge the batch instead)")
        Fn.join_side_stream()          # weight gradients may still be in flight on the side stream
        a = self.arena
        if a.flat_p.is_cuda:
            ops.flush_reductions()     # bias / LayerNorm gradients: one batched final reduction (no-op when already flushed)
        self._adopt_stray_grads()
        # tensors without a gradient this step (norm3.*, object_model.norm.*, ...) must not be updated: their slices read zero
        a.zero_untouched(lambda i: a.params[i].grad is not None)
        self._ever.update(i for i in range(len(a.params)) if i not in a._clean)

    def _sync_hyper(self, grad_scale):
        g = self.param_groups[0]
        want = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), float(grad_scale))
        if self._hyper is None:
            self._hyper = torch.zeros(8, device=self.arena.flat_p.device, dtype=torch.float32)
            self._hyper_host = None
        if want != self._hyper_host:
            self._hyper[:6].copy_(torch.tensor(want, dtype=torch.float32))
            self._hyper_host = want
        if self._hyper_step != self.step_count:                 # after load_state_dict: the device counter follows the host's
            self._hyper[6:7].fill_(float(self.step_count))
            self._hyper_step = self.step_count

    # ---- update spread over the backward pass ---------------------------------------------------------------------
    def begin_overlapped(self, grad_scale=1.0):
        """Call after zero_grad() and before the step's forward / backward (no gradient exchange between ranks in this step): the
        device step counter advances now, and every transformer layer's weights -- 71 % of the parameters -- are updated on the
        gradient side stream at the end of that layer's backward (functional._early_update), where the HBM-bound update runs
        beside the MFMA-bound backward; launch() then only covers what is left (embeddings, heads, the vector tail).  Element for
        element the same arithmetic as the single launch."""
        if not self.arena.flat_p.is_cuda:
            return
        self._sync_hyper(grad_scale)
        ops.adamw_prep_dev(self._hyper)
        self._early = []
        self._pending = {}
        Fn.EARLY_OPT = self._early_update
        Fn.TAKE_PENDING_OPT = self._take_pending

    def abort_overlapped(self):
        """Drop an overlapped step that did not reach launch() (forward / backward raised): clears the per-layer hook and the record
        of early updates so the next zero_grad() / step starts clean.  Layers already updated stay updated (their moments too)."""
        if Fn.EARLY_OPT is not None and getattr(Fn.EARLY_OPT, "__self__", None) is self:
            Fn.EARLY_OPT = None
            Fn.TAKE_PENDING_OPT = None
        self._early = None
        self._pending = {}
        # begin_overlapped() already advanced the DEVICE step counter (adamw_prep_dev) while the host's step_count stays: make the next
        # _sync_hyper() rewrite the device counter from the host's, or every later step would run its bias correction one step ahead
        self._hyper_step = -1

    def _early_update(self, params, now=False):
        """``now``: launch the update here instead of leaving it pending for the next weight-gradient group of this stream."""
        a = self.arena
        idx = sorted(self._index_of[id(q)] for q in params if id(q) in self._index_of)
        runs = []
        for i in idx:                                           # adjacent slices (a layer's matrices are) merge into one launch
            lo, hi = a.offsets[i], a.offsets[i] + (a.params[i].numel() + a.ALIGN - 1) // a.ALIGN * a.ALIGN
            if runs and runs[-1][1] == lo:
                runs[-1][1] = hi
            else:
                runs.append([lo, hi])
        # Round 6: the layer's (contiguous) range is not launched here but left PENDING for the next grouped weight-gradient launch on this
        # stream -- the layer below's, a few kernels later -- whose 40 spare workgroups (216 GEMM workgroups on 256 CUs) run it beside the
        # products (dvlp_wgrad_grouped_ex) instead of as a 35-us launch of its own between two of them.  Per stream: the text tower's
        # backward runs on its own stream when the towers are concurrent, and an update may only ride in a launch that is ordered behind
        # the kernels that finished its gradients.
        sid = torch.cuda.current_stream().cuda_stream
        self._flush_pending(sid)                                 # (one nobody took: two layers without a weight-gradient group in between)
        for lo, hi in (runs if now else runs[:-1]):
            ops.adamw_range_dev(a.flat_p, a.flat_g, self.m, self.v, self._hyper, a.flat_s, lo, hi)
        if runs and not now:
            self._pending[sid] = tuple(runs[-1])
        for lo, hi in runs:
            self._early.append((lo, hi))

    def _take_pending(self):
        """(p, g, m, v, hyper, shadow, lo, hi) of the update pending on the CURRENT stream (then no longer pending), or None."""
        pend = getattr(self, "_pending", None)
        if not pend:
            return None
        r = pend.pop(torch.cuda.current_stream().cuda_stream, None)
        if r is None:
            return None
        a = self.arena
        return (a.flat_p, a.flat_g, self.m, self.v, self._hyper, a.flat_s, r[0], r[1])

    def _flush_pending(self, sid=None):
        """Launch pending updates on their own (``sid``: only that stream's; None: all -- the caller has joined the streams)."""
        pend = getattr(self, "_pending", None)
        if not pend:
            return
        a = self.arena
        for k in [k for k in pend if sid is None or k == sid]:
            lo, hi = pend.pop(k)
            ops.adamw_range_dev(a.flat_p, a.flat_g, self.m, self.v, self._hyper, a.flat_s, lo, hi)

    def _finish_overlapped(self):
        """The ranges no layer updated, as few launches as the arena layout allows."""
        a = self.arena
        Fn.EARLY_OPT = None
        Fn.TAKE_PENDING_OPT = None
        self._flush_pending()
        done, self._early = sorted(self._early), None
        pos = 0
        for lo, hi in done + [(a.total, a.total)]:
            if lo > pos:
                ops.adamw_range_dev(a.flat_p, a.flat_g, self.m, self.v, self._hyper, a.flat_s, pos, lo)
            pos = max(pos, hi)

    def launch(self, grad_scale=1.0):
        """Device-side part: the fused update (hyper-parameters and the step counter live in device memory, so this launch can sit
        inside a captured hipGraph and be replayed)."""
        a = self.arena
        if self._early is not None:
            if float(grad_scale) != self._hyper_host[5]:
                raise RuntimeError("FusedAdamW: grad_scale changed between begin_overlapped() and launch()")
            self._finish_overlapped()
        else:
            self._sync_hyper(grad_scale)
            ops.adamw_step_dev(a.flat_p, a.flat_g, self.m, self.v, self._hyper, a.flat_s)
        self.step_count += 1
        self._hyper_step = self.step_count
        if a.flat_s is not None:
            a.adopt_shadow()
        else:
            Fn.SHADOWS.invalidate()    # the masters moved without a version bump: bf16 shadows cast on demand are stale now

    def replayed(self):
        """A captured graph containing launch() was replayed: the device advanced its step counter, follow it on the host."""
        self.step_count += 1
        self._hyper_step = self.step_count

    def step(self, grad_scale=1.0):
        self.prepare()
        self.launch(grad_scale)

    # ---- checkpoint interchange (base/base_trainer.py:176-267) ----------------------------------------------------
    def state_dict(self):
        a = self.arena
        state = {}
        for k, i in enumerate(a.model_order):
            if i in self._ever:
                lo, hi = a.slice_of(i)
                shape = a.params[i].shape
                state[k] = dict(step=self.step_count, exp_avg=self.m[lo:hi].view(shape).clone(), exp_avg_sq=self.v[lo:hi].view(shape).clone())
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(a.model_order)))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        a = self.arena
        self.m.zero_()
        self.v.zero_()
        self._ever = set()
        steps = set()
        for k, st in sd["state"].items():
            i = a.model_order[int(k)]
            lo, hi = a.slice_of(i)
            self.m[lo:hi].copy_(st["exp_avg"].reshape(-1))
            self.v[lo:hi].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
            self._ever.add(i)
        if len(steps) > 1:
            raise NotImplementedError(f"per-parameter step counts differ ({sorted(steps)}): the fused kernel keeps one step counter")
        self.step_count = steps.pop() if steps else 0
        g = sd["param_groups"][0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in g:
                self.param_groups[0][k] = tuple(g[k]) if k == "betas" else g[k]


def adjust_learning_rate(optimizer, epoch, args):
    """trainer/trainer_dist.py:97-102, called at the END of every epoch (:198): every group's lr becomes ``args.learning_rate1``
    (default 2e-4, train_dist_multi.py:173) x 0.1 per milestone of ``args.schedule`` already passed -- so epoch 1 runs at the
    config's lr and every later epoch at learning_rate1 unless -lr1 is given.  Reproduced verbatim, quirk included."""
    lr = args.learning_rate1
    for milestone in args.schedule:
        lr *= 0.1 if epoch >= milestone else 1.0
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr
    return lr


def save_checkpoint(path, model, optimizer, epoch, monitor_best=0.0, config=None):
    """The reference's checkpoint file (base/base_trainer.py:185-192): same keys, same nesting."""
    torch.save({"arch": type(model).__name__, "epoch": epoch, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict(),
                "monitor_best": monitor_best, "config": config}, path)


def resume_checkpoint(path, model, optimizer, map_location="cpu"):
    """base/base_trainer.py:202-267: model weights (``module.`` prefix added / stripped as needed) and optimizer state.
    Returns (start_epoch, monitor_best)."""
    from .model import state_dict_data_parallel_fix
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(state_dict_data_parallel_fix(ck["state_dict"], model.state_dict()))
    if isinstance(optimizer, FusedAdamW) and optimizer.arena.flat_s is not None:
        optimizer.arena.refresh_shadow()
    else:
        Fn.SHADOWS.invalidate()
    optimizer.load_state_dict(ck["optimizer"])
    return ck["epoch"] + 1, ck["monitor_best"]


class GradReducer:
    """Bucketed gradient all-reduce (sum) over torch.distributed (RCCL on MI355X, gloo in the CPU tests).

    Buckets are contiguous ranges of the arena's flat gradient buffer, so a bucket is reduced in place with one
    collective.  A bucket is launched (async, on the communication stream) as soon as every parameter in it that is
    known to receive a gradient has fired its post-accumulate hook; ``finish()`` launches the rest and waits."""

    def __init__(self, arena: ParamArena, bucket_mb: float = 64.0, group=None, always_reduce: bool = False):
        self.arena = arena
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # always_reduce: issue the collectives even in a one-rank group (exercises the RCCL path on a single GPU)
        self.collective = self.world > 1 or (always_reduce and dist.is_initialized())
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []          # (lo, hi, [param indices])
        lo, idxs = 0, []
        for i, p in enumerate(arena.params):
            idxs.append(i)
            hi = arena.offsets[i] + (p.numel() + arena.ALIGN - 1) // arena.ALIGN * arena.ALIGN
            # always cut at the matrix / vector boundary: vector gradients are only final after the batched reduction
            if hi - lo >= cap or i + 1 == arena.n_matrix:
                self.buckets.append((lo, hi, idxs))
                lo, idxs = hi, []
        if idxs:
            self.buckets.append((lo, arena.total, idxs))
        self.tail = set(b for b, (blo, _, _) in enumerate(self.buckets) if blo >= arena.vec_lo)   # reduced in finish() only
        self.bucket_of = {}
        for b, (_, _, idxs) in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = b
        if arena.flat_g.is_cuda:
            ops.enable_deferred_reductions(arena.flat_g.device)
        self._issue_stream = None
        self.expected = None       # per bucket: set of param indices known to get gradients (learned on step 1)
        self._seen = set()
        self._pending = None
        self._handles = []
        self._launched = set()
        for i, p in enumerate(arena.params):
            if p.requires_grad:
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(_p):
            self._seen.add(i)
            if self.expected is None or not self.collective:
                return
            b = self.bucket_of[i]
            self._pending[b].discard(i)
            if not self._pending[b] and b not in self._launched and b not in self.tail:
                self._launch(b)
        return hook

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self._launched.add(b)
        # with weight gradients on the side stream, the collective must be ordered after BOTH streams: issue it from the
        # side stream after making that wait for the main stream (RCCL's own stream then waits for the side stream)
        if Fn.OVERLAP_WGRAD and self.arena.flat_g.is_cuda:
            # issue from a third stream that waits for both compute streams: neither of them stalls behind the other (making
            # the side stream wait for the main one here serialised the weight-gradient GEMMs it exists to overlap)
            if self._issue_stream is None:
                self._issue_stream = torch.cuda.Stream(device=self.arena.flat_g.device)
            cs = self._issue_stream
            cs.wait_stream(torch.cuda.current_stream())
            cs.wait_stream(ops.side_stream())
            with torch.cuda.stream(cs):
                h = dist.all_reduce(self.arena.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            h = dist.all_reduce(self.arena.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._handles.append(h)

    def begin(self):
        self._seen = set()
        self._handles, self._launched = [], set()
        if self.expected is not None:
            self._pending = [set(e) for e in self.expected]

    def finish(self):
        """Call after backward: zero never-touched slices, reduce what is left, wait for everything."""
        Fn.join_side_stream()
        a = self.arena
        if a.flat_g.is_cuda:
            ops.flush_reductions()         # finishes the vector gradients of the tail bucket(s)
        a.zero_untouched(lambda i: i in self._seen)
        if self.collective:
            for b in range(len(self.buckets)):
                if b not in self._launched:
                    self._launch(b)
            for h in self._handles:
                h.wait()
        if self.expected is None:
            self.expected = [set(i for i in idxs if i in self._seen) for (_, _, idxs) in self.buckets]
        return 1.0 / self.world          # fold the DDP average into the optimizer's grad_scale


def _gather_plain(t, world, group=None):
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t.contiguous(), group=group)
    return torch.cat(out, 0)


def gather_embeddings(out, text_length, text_mask, args):
    """Cross-GPU negatives (opt-in; the reference binds ``self.allgather = AllGather_multi.apply`` at trainer_dist.py:82 and never
    calls it in ``_train_epoch``): every rank sees the embeddings of all ranks, concatenated in rank order, with the reference's
    backward (the local slice of the incoming gradient, no reduce-scatter -- trainer_dist.py:25-31).  Masks and lengths carry no
    gradient and are gathered plainly."""
    n_gpu = args.world_size
    g = lambda t: AllGather_multi.apply(t, n_gpu, args)  # noqa: E731
    gathered = dict(global_text_embeddings=g(out["global_text_embeddings"]), local_text_embeddings=g(out["local_text_embeddings"]),
                    global_object_embeddings=g(out["global_object_embeddings"]), local_object_embeddings=g(out["local_object_embeddings"]),
                    object_mask=_gather_plain(out["object_mask"], n_gpu))
    return gathered, _gather_plain(text_length, n_gpu), _gather_plain(text_mask, n_gpu)


class _AdoptGathered(torch.autograd.Function):
    """``AllGather_multi`` with the collective ALREADY done: forward hands out the buffer every rank's slice was gathered into, backward
    returns the local slice of its gradient (trainer/trainer_dist.py:25-31).  Launches nothing -- what lets ``GraphedTrainStep`` keep the
    all-gather outside its hipGraphs: the forward graph ends at the local embeddings, the collective runs between two replays, and the
    loss graph starts from the gathered buffers."""

    @staticmethod
    def forward(ctx, local, gathered, rank):
        ctx.rank, ctx.batch_size = rank, local.shape[0]
        return gathered.view(gathered.shape)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output[ctx.batch_size * ctx.rank:ctx.batch_size * (ctx.rank + 1)], None, None


GATHERED_KEYS = ("global_text_embeddings", "local_text_embeddings", "global_object_embeddings", "local_object_embeddings", "object_mask")


def forward_only(model, data):
    """trainer/trainer_dist.py:148-159: the model's forward and the caption-side inputs of the loss.  -> (out dict, text_length, text_mask)"""
    att = data["text"]["attention_mask"]
    out = model(data)
    if att.is_cuda and att.dtype == torch.int64 and att.dim() == 2 and att.shape[1] >= 2:
        text_length, text_mask = ops.text_mask_len(att)          # one launch instead of a reduce, a slice copy, a subtraction and a multiplication
    else:
        text_length = torch.sum(att, dim=1)
        text_mask = (att[:, 1:] - 1.0) * 100.0                   # (:156-159; the subtraction already yields a contiguous tensor)
    return out, text_length, text_mask


def loss_backward_first(model, loss_fn, out, text_length, text_mask):
    """trainer/trainer_dist.py:160-165: sim_matrix, GlobalLocalLoss and ``loss.backward()`` (down to the object tower's topmost open cut)."""
    global_sim = sim_matrix(out["global_text_embeddings"], out["global_object_embeddings"])
    loss, global_loss, local_loss = loss_fn(global_sim, out["local_object_embeddings"], out["local_text_embeddings"],
                                            out["object_mask"], text_length, text_mask)
    loss.backward()
    if loss.is_cuda:
        Fn.join_side_stream()          # weight gradients (side stream) and the text tower's stream
    return loss.detach(), global_loss.detach(), local_loss.detach()


def backward_first(model, loss_fn, data, gather_negatives=None):
    """trainer/trainer_dist.py:148-165: forward, masks, sim_matrix, GlobalLocalLoss and ``loss.backward()``.  When the object tower
    carries a ``grad_cut`` the backward stops at that block's input (text tower, heads and the upper object blocks are done; their
    gradients are final when this returns) and ``backward_second`` finishes it.  Returns the three detached losses."""
    out, text_length, text_mask = forward_only(model, data)
    if gather_negatives is not None:
        out, text_length, text_mask = gather_embeddings(out, text_length, text_mask, gather_negatives)
    return loss_backward_first(model, loss_fn, out, text_length, text_mask)


def backward_next(model):
    """Resume a cut backward by ONE piece: from the topmost open cut down to the next one (or to the inputs).  False when no cut is open."""
    om = getattr(model, "object_model", None)
    cut = om.take_cut() if om is not None and hasattr(om, "take_cut") else None
    if cut is None:
        return False
    tok, leaf = cut
    torch.autograd.backward(tok, leaf.grad)
    return True


def finish_backward(model):
    """The batched final reduction of the bias / LayerNorm gradients (after the last piece of the backward)."""
    if torch.cuda.is_available() and next(model.parameters()).is_cuda:
        Fn.join_side_stream()          # deferred partial sums may have been produced on the side stream
        ops.flush_reductions()


def backward_second(model):
    """The rest of a cut backward (every open piece; no-op without a cut) + the batched final reduction."""
    while backward_next(model):
        pass
    finish_backward(model)


def forward_backward(model, loss_fn, data, gather_negatives=None):
    """One forward + backward (both pieces when the object tower carries a ``grad_cut``).  Returns the three detached losses."""
    losses = backward_first(model, loss_fn, data, gather_negatives)
    backward_second(model)
    return losses


def train_step(model, loss_fn, optimizer, data, reducer: GradReducer | None = None, gather_negatives=None):
    """One optimisation step on an already-tokenised, already-on-device batch (trainer/trainer_dist.py:144-171).
    ``gather_negatives``: None (reference behaviour: per-rank negatives) or an object with ``world_size`` / ``rank`` (the
    reference's ``args``) to run the contrastive losses over the all-gathered embeddings of every rank."""
    optimizer.zero_grad()
    if reducer is not None:
        reducer.begin()
    elif isinstance(optimizer, FusedAdamW) and not (dist.is_initialized() and dist.get_world_size() > 1):
        optimizer.begin_overlapped()           # no gradient exchange: the layers' updates ride along with the backward
    try:
        losses = forward_backward(model, loss_fn, data, gather_negatives)
        scale = reducer.finish() if reducer is not None else 1.0
    except BaseException:
        if isinstance(optimizer, FusedAdamW):
            optimizer.abort_overlapped()         # a bad batch must not wedge the optimizer ("begin_overlapped() without ...")
        raise
    if isinstance(optimizer, FusedAdamW):
        optimizer.step(grad_scale=scale)
    else:
        if scale != 1.0:
            for p in model.parameters():
                if p.grad is not None:
                    p.grad.mul_(scale)
        optimizer.step()
    return losses


class GraphedTrainStep:
    """``train_step`` as hipGraph replays: after ``warmup`` eager steps (kernel attributes set, workspaces allocated, the deferred-
    reduction table uploaded, the grad-less tensors learnt) the next step is captured -- forward, losses, backward, the batched
    final reduction, fused AdamW: ~520 launches -- and every later call copies the batch into the captured input buffers and
    replays.  The host's per-launch Python / ctypes / autograd cost (~40 us x 520) disappears from the step.

    Data parallel (``world > 1``): no collective is captured.  The backward is cut at the object blocks ``cut`` (an index or several,
    default (8, 4)) and captured as ``len(cut) + 1`` graphs sharing one memory pool.  When piece k has run, the weight gradients it
    produced are final -- piece 0: the text tower and the object blocks above the first cut (contiguous arena runs), piece k: the
    blocks between two cuts, last piece: the lowest blocks, the prologue and the vector tail -- and their exchange starts at once:
    every run is all-reduced in ``bucket_mb`` pieces on the communication stream while the NEXT graph runs, and each piece is
    handed to the fused optimizer on a third stream the moment its reduction completes (``adamw_range_dev``; the 1 / world average
    is folded into its grad_scale), so neither the optimizer nor any exchange but the last piece's (blocks 0-3 + prologue: ~20 % of
    the bytes at the default cuts) waits for the whole arena.  One captured set per batch shape (up to ``max_shapes``): a batch of another
    shape -- a loader's smaller last batch -- runs ``warmup`` eager steps of its own, is captured, and from then on both shapes replay."""

    def __init__(self, model, loss_fn, optimizer: FusedAdamW, warmup: int = 2, group=None, bucket_mb: float = 64.0, always_reduce: bool = False,
                 cut=(8, 4), grad_dtype: str = "float32", time_exchange: bool = False, exchange: str = "all_reduce", gather_negatives=None):
        """``grad_dtype='bfloat16'``: every bucket crosses the links as bf16 (half the bytes: 292 instead of 584 MB per step) -- cast by a
        kernel into a staging buffer, summed by the collective in bf16, cast back into the fp32 gradient arena, where the optimizer reads
        it; moments and master weights stay fp32.  Two extra HBM passes per bucket (12 B per parameter beside the update's 30).
        ``time_exchange``: HIP events around every piece's collectives on the communication stream (``exchange_times()``).
        ``exchange``: ``'all_reduce'`` (default: RCCL's own choice of algorithm, a ring on most topologies) or ``'rs_ag'`` -- every bucket as an
        in-place ``reduce_scatter_tensor`` (each rank ends up owning 1/world of the bucket's sum) followed by an in-place
        ``all_gather_into_tensor``: the two DIRECT collectives that put all seven xGMI links of a GPU to work at once instead of one ring
        hop at a time (0.18 against 1.3 ms for the exposed last piece at 8 GPUs, on paper).  With ``grad_dtype='bfloat16'``
        the scatter stays fp32 and only the gather crosses as bf16 (one rounding, whatever the world size; 6 instead of 4 bytes per element
        on the links).  NCCL / RCCL groups only; other backends (gloo in the CPU tests) take the all-reduce path.  Unmeasured on > 1 GPU.
        ``gather_negatives``: None (the reference's training: per-rank negatives) or an object with ``world_size`` / ``rank`` (the reference's
        ``args``): the contrastive losses run over the all-gathered embeddings of every rank with ``AllGather_multi``'s backward (the local
        slice, trainer/trainer_dist.py:13-31).  The step is then captured as a FORWARD graph (both towers, up to the local embeddings) and the
        loss / backward graphs; the all-gather runs between them, outside any capture, into persistent buffers the loss graph reads."""
        if grad_dtype not in ("float32", "bfloat16"):
            raise ValueError("grad_dtype must be 'float32' or 'bfloat16'")
        if exchange not in ("all_reduce", "rs_ag"):
            raise ValueError("exchange must be 'all_reduce' or 'rs_ag'")
        self.exchange = exchange
        self.gather, self._gbuf, self._gbufs, self._fwd = gather_negatives, None, {}, None
        self.exchange_used, self._warned = set(), False       # the forms actually taken (bench.py reports them), one warning on an emulated one
        self.grad_dtype, self.time_exchange, self._xev, self._stage = grad_dtype, bool(time_exchange), [], None
        self.model, self.loss_fn, self.opt = model, loss_fn, optimizer
        self.warmup, self.calls = max(1, warmup), 0
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # (a multiple of the arena's alignment: adamw_range_dev wants 16-byte aligned range starts, and an odd bucket size would make
        #  every second piece fail in the middle of a step)
        al = optimizer.arena.ALIGN
        self.bucket = max(al, int(bucket_mb * 1024 * 1024 / 4) // al * al)
        # always_reduce: issue the collectives even in a one-rank group (exercises the RCCL path on a single GPU)
        self.collective = self.world > 1 or (always_reduce and dist.is_initialized())
        self.graphs, self.static, self.out, self.shape_key = None, None, None, None
        # one captured set per batch shape (a loader's last batch is smaller; evaluation-sized batches come and go): shape key ->
        # (graphs, input buffers, output tensors), plus the calls seen per shape -- every shape gets its own eager warm-up before its capture
        self._sets, self._seen, self.max_shapes = {}, {}, 4
        self.cuts, self.piece_runs, self._comm, self._optst = (), [[(0, optimizer.arena.total)]], None, None
        om = getattr(model, "object_model", None)
        if self.collective and cut and om is not None and hasattr(om, "grad_cut"):
            cuts = sorted({int(c) for c in ((cut,) if isinstance(cut, int) else cut) if 0 < int(c) < len(om.blocks)}, reverse=True)
            if cuts:
                self.cuts = tuple(cuts)
                self.piece_runs = self.plan_exchange(optimizer.arena, self.cuts)

    # kept for callers / tests of the two-piece form
    @property
    def cut(self):
        return self.cuts[0] if self.cuts else None

    @property
    def graph(self):
        return self.graphs[0] if self.graphs else None

    @property
    def graph2(self):
        return self.graphs[1] if self.graphs and len(self.graphs) > 1 else None

    @property
    def early_runs(self):
        return [r for runs in self.piece_runs[:-1] for r in runs]

    @property
    def late_runs(self):
        return self.piece_runs[-1]

    @staticmethod
    def plan_exchange(arena, cuts):
        """Arena ranges whose gradients are final after each piece of a backward cut at object blocks ``cuts`` (descending): piece 0 =
        the text tower's matrices and the object blocks >= cuts[0]; piece k = blocks in [cuts[k], cuts[k-1]); the last piece = the
        complement (lowest blocks, prologue, heads written late, the vector tail).  Every element of the arena is in exactly one piece."""
        cuts = (cuts,) if isinstance(cuts, int) else tuple(cuts)
        bounds = [len(cuts) and 10 ** 9] + list(cuts)

        def piece_of(n):
            if n.startswith("text_model."):
                return 0
            if n.startswith("object_model.blocks."):
                blk = int(n.split(".")[2])
                for k in range(len(cuts)):
                    if cuts[k] <= blk < bounds[k]:
                        return k
            return len(cuts)
        pieces = [[] for _ in range(len(cuts) + 1)]
        for i in range(arena.n_matrix):
            k = piece_of(arena.names[i])
            if k == len(cuts):
                continue
            lo = arena.offsets[i]
            hi = lo + (arena.params[i].numel() + arena.ALIGN - 1) // arena.ALIGN * arena.ALIGN
            if pieces[k] and pieces[k][-1][1] == lo:
                pieces[k][-1][1] = hi
            else:
                pieces[k].append([lo, hi])
        early = sorted((lo, hi, k) for k in range(len(cuts)) for lo, hi in pieces[k] if hi - lo >= (1 << 18))   # tiny runs ride with the last piece
        out = [[] for _ in range(len(cuts) + 1)]
        pos = 0
        for lo, hi, k in early + [(arena.total, arena.total, len(cuts))]:
            if lo > pos:
                out[len(cuts)].append((pos, lo))
            if hi > lo:
                out[k].append((lo, hi))
            pos = hi
        return out

    @staticmethod
    def _key(data):
        return tuple((tuple(t.shape), t.dtype) for t in (data["text"]["input_ids"], data["text"]["attention_mask"], data["object"], data["object_mask"]))

    def _pieces(self, runs):
        for lo, hi in runs:
            for p in range(lo, hi, self.bucket):
                yield p, min(hi, p + self.bucket)

    def _set_cut(self, on=True):
        # only for this object's own forward: a caller that runs ``loss.backward()`` itself must get the whole backward
        if self.cuts:
            self.model.object_model.grad_cut = self.cuts if on else None

    # ---- the step as a list of pieces: piece 0 = forward + losses + the top of the backward, piece k = the next stretch of it --------
    def _piece(self, k, data):
        """Piece k of the step.  With ``gather_negatives`` the pieces are shifted by one: piece -1... is numbered 0 = the forward alone
        (``n_pieces`` = cuts + 2), the all-gather follows it outside the graphs (``_gather``), piece 1 = losses + the top of the backward."""
        g = 1 if self.gather is not None else 0
        last = k == len(self.cuts) + g
        if k == 0:
            self.opt.zero_grad()
            if not self.collective:
                self.opt.begin_overlapped()
            self._set_cut()
            try:
                if g:
                    self._fwd = forward_only(self.model, data)
                else:
                    self.out = backward_first(self.model, self.loss_fn, data)
            except BaseException:
                self.opt.abort_overlapped()
                raise
            finally:
                self._set_cut(False)
            if not g:
                self.opt._adopt_stray_grads()
            else:
                return
        elif g and k == 1:
            out, tl, tm = self._fwd
            rank = self.gather.rank
            gathered = {key: (_AdoptGathered.apply(out[key], self._gbuf[key], rank) if out[key].requires_grad else self._gbuf[key]) for key in GATHERED_KEYS}
            try:
                self.out = loss_backward_first(self.model, self.loss_fn, gathered, self._gbuf["text_length"], self._gbuf["text_mask"])
            except BaseException:
                self.opt.abort_overlapped()
                raise
            finally:
                # keep the forward's OUTPUT BUFFERS (the all-gather of the next replay reads them), not its autograd graph: a graph kept
                # alive across steps keeps the parameters' AccumulateGrad nodes alive with the stream they were created on (the eager
                # warm-up's), and the engine then ties that stream to the capture stream of the next capture -- the capture never ends
                del gathered
                self._fwd = ({key: t.detach() for key, t in out.items()}, tl.detach(), tm.detach())
            self.opt._adopt_stray_grads()
        else:
            backward_next(self.model)
            if not last:
                # this piece's gradients are exchanged (and consumed by AdamW) right after it: weight gradients still in flight on the
                # gradient side stream (Fn.OVERLAP_WGRAD) must be joined and stray p.grad copies adopted HERE, not only in the last piece
                Fn.join_side_stream()
                self.opt._adopt_stray_grads()
        if last:
            finish_backward(self.model)
            self.opt.prepare()
            if not self.collective:
                self.opt.launch(grad_scale=1.0)

    def _gather_buffers(self):
        """The persistent gathered buffers for the shapes of the last forward (one set per shape, kept for as long as this object lives: a
        captured loss graph points at them).  Allocates outside captures only."""
        out, tl, tm = self._fwd
        world = self.gather.world_size
        local = dict(out)
        local["text_length"], local["text_mask"] = tl, tm
        key = tuple((k, tuple(v.shape), v.dtype) for k, v in local.items())
        buf = self._gbufs.get(key)
        if buf is None:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("gather buffers would be allocated inside a hipGraph capture")
            buf = {k: torch.empty((world * v.shape[0],) + tuple(v.shape[1:]), device=v.device, dtype=v.dtype) for k, v in local.items()}
            self._gbufs[key] = buf
        self._gbuf = buf
        return local, buf

    def _gather(self):
        """Between the forward graph and the loss graph, outside any capture: every rank's embeddings, masks and caption lengths into the
        persistent buffers the loss piece adopts (rank order along dim 0, as ``torch.cat(all_gather(...))`` -- trainer_dist.py:17-23)."""
        local, buf = self._gather_buffers()
        world = self.gather.world_size
        if dist.is_initialized() and world > 1:
            for k, v in local.items():
                dist.all_gather(list(buf[k].chunk(world, 0)), v.detach().contiguous(), group=self.group)
        else:
            for k, v in local.items():
                buf[k].copy_(v.detach())

    def _exchange_and_update(self, runs, piece=0):
        """All-reduce the given arena ranges of the gradient buffer (sum) and update the parameters of each reduced piece: collectives
        queue on the communication stream behind what the current stream holds so far, each fused-AdamW range launch waits (on the
        optimizer stream) for just its own piece."""
        a = self.opt.arena
        g = a.flat_g
        pieces = list(self._pieces(runs))
        if not pieces:
            return
        half = self.grad_dtype == "bfloat16"
        # exchange='rs_ag': every bucket's prefix that divides by the world size goes as reduce-scatter + all-gather, the < world elements
        # left over (world sizes that do not divide the 64-element bucket alignment: 3, 6, ...) as a small all-reduce -- the form taken no
        # longer depends on the bucket boundaries.  NCCL / RCCL run the two collectives natively and in place; other backends (gloo: the
        # CPU and two-ranks-on-one-GPU tests) take `_rs_ag_emulated`, which walks the same shard views in the same order.
        direct = self.exchange == "rs_ag"
        native = direct and dist.get_backend(self.group) == "nccl"
        self.exchange_used.add("rs_ag" if native else "rs_ag (emulated: all_reduce + all_gather)" if direct else "all_reduce")
        if direct and not native and not self._warned:
            self._warned = True
            warnings.warn("GraphedTrainStep(exchange='rs_ag'): backend %r has no reduce_scatter_tensor; the shard walk is emulated with all_reduce + "
                          "all_gather (same result, no link-level benefit)" % dist.get_backend(self.group))
        if not g.is_cuda:
            for lo, hi in pieces:
                if direct:
                    self._rs_ag_host(g, lo, hi, half)
                elif half:
                    h16 = g[lo:hi].to(torch.bfloat16)
                    dist.all_reduce(h16, op=dist.ReduceOp.SUM, group=self.group)
                    g[lo:hi].copy_(h16)
                else:
                    dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            return
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=g.device)
            self._optst = torch.cuda.Stream(device=g.device)
        if half and self._stage is None:
            self._stage = torch.empty(a.total, device=g.device, dtype=torch.bfloat16)      # one slot per arena element: buckets never share staging
        self._comm.wait_stream(torch.cuda.current_stream())
        ev = None
        with torch.cuda.stream(self._comm):
            if self.time_exchange:
                # bytes per element that cross the links, up to the (world - 1) / world factor: all-reduce 2 x (4 | 2), scatter 4 + gather (4 | 2)
                per = (4 + (2 if half else 4)) if direct else 2 * (2 if half else 4)
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), sum(hi - lo for lo, hi in pieces) * per // 2)
                ev[0].record()
            if direct:
                # reduce-scatter + all-gather, both in place (RCCL's in-place forms: the output shard IS the rank's slice of the input, the
                # gathered buffer starts where the shards lie); same stream, so the gather is ordered behind the scatter
                r = dist.get_rank(self.group)
                handles = []
                for lo, hi in pieces:
                    mid, sh = self.shard_split(lo, hi, self.world)
                    hs = []
                    if sh:
                        mine = g[lo + r * sh: lo + (r + 1) * sh]
                        if native:
                            hrs = dist.reduce_scatter_tensor(mine, g[lo:mid], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                        else:
                            hrs = self._reduce_scatter_emulated(mine, g[lo:mid], r, sh)
                        if half:
                            hrs.wait()                                                  # the cast below runs on THIS stream, not on the collective's
                            mine16 = self._stage[lo + r * sh: lo + (r + 1) * sh]
                            ops.cast(mine, torch.bfloat16, out=mine16)                  # the owned shard's fp32 sum, rounded once
                            hs.append(dist.all_gather_into_tensor(self._stage[lo:mid], mine16, group=self.group, async_op=True))
                        else:
                            hs.append(dist.all_gather_into_tensor(g[lo:mid], mine, group=self.group, async_op=True))
                    if mid < hi:                                                        # the remainder: fewer than `world` elements
                        if half:
                            ops.cast(g[mid:hi], torch.bfloat16, out=self._stage[mid:hi])
                            hs.append(dist.all_reduce(self._stage[mid:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                        else:
                            hs.append(dist.all_reduce(g[mid:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    handles.append(hs)
            elif half:
                for lo, hi in pieces:
                    ops.cast(g[lo:hi], torch.bfloat16, out=self._stage[lo:hi])
                handles = [[dist.all_reduce(self._stage[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)] for lo, hi in pieces]
            else:
                handles = [[dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)] for lo, hi in pieces]
            if ev is not None:
                for hs in handles:
                    for h in hs:
                        h.wait()
                ev[1].record()
                self._xev.append(ev + (piece,))
        with torch.cuda.stream(self._optst):
            for hs, (lo, hi) in zip(handles, pieces):
                for h in hs:
                    h.wait()                                  # this stream waits for that bucket's collectives only
                if half:
                    ops.cast(self._stage[lo:hi], torch.float32, out=g[lo:hi])
                ops.adamw_range_dev(a.flat_p, g, self.opt.m, self.opt.v, self.opt._hyper, a.flat_s, lo, hi)

    @staticmethod
    def shard_split(lo, hi, world):
        """(mid, shard): [lo, mid) is the prefix of the bucket that divides by ``world`` (rank r owns [lo + r shard, lo + (r + 1) shard)),
        [mid, hi) the remainder that goes as a plain all-reduce."""
        sh = (hi - lo) // world
        return lo + sh * world, sh

    def _reduce_scatter_emulated(self, mine, whole, r, sh):
        """Backends without reduce_scatter_tensor: the sum of ``whole`` over the ranks, of which only this rank's shard is kept (the other
        shards of ``whole`` are left as the native in-place form leaves them: not to be read before the gather)."""
        tmp = whole.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
        mine.copy_(tmp[r * sh:(r + 1) * sh])
        whole[:r * sh].fill_(float("nan"))                    # poison what the gather must overwrite: a wrong shard offset shows up as NaN
        whole[(r + 1) * sh:].fill_(float("nan"))

        class _Done:
            def wait(self):
                return True
        return _Done()

    def _rs_ag_host(self, g, lo, hi, half):
        """The rs_ag walk on host tensors (gloo, CPU tests): same shard views and order as the device path."""
        r = dist.get_rank(self.group)
        mid, sh = self.shard_split(lo, hi, self.world)
        if sh:
            mine = g[lo + r * sh: lo + (r + 1) * sh]
            self._reduce_scatter_emulated(mine, g[lo:mid], r, sh)
            if half:
                mine16 = mine.to(torch.bfloat16)
                out16 = torch.empty(mid - lo, dtype=torch.bfloat16)
                dist.all_gather_into_tensor(out16, mine16, group=self.group)
                g[lo:mid].copy_(out16)
            else:
                out = torch.empty(mid - lo, dtype=g.dtype)
                dist.all_gather_into_tensor(out, mine.clone(), group=self.group)
                g[lo:mid].copy_(out)
        if mid < hi:
            if half:
                h16 = g[mid:hi].to(torch.bfloat16)
                dist.all_reduce(h16, op=dist.ReduceOp.SUM, group=self.group)
                g[mid:hi].copy_(h16)
            else:
                dist.all_reduce(g[mid:hi], op=dist.ReduceOp.SUM, group=self.group)

    def exchange_times(self):
        """[(milliseconds, bytes, piece)] per exchanged piece since the last call, in issue order (``time_exchange=True``): the time the
        piece's collectives occupied the communication stream -- NOT what the step waited for (all but the last piece run beside the next
        graph).  A piece with nothing to exchange records no entry: group by the piece index, not by position."""
        torch.cuda.synchronize()
        out = [(a.elapsed_time(b), n, k) for a, b, n, k in self._xev]
        self._xev = []
        return out

    def _begin_updates(self):
        """Per step, before the first range update: hyper-parameters to the device if they changed, the device step counter advanced --
        on the optimizer stream, behind the previous step's last update."""
        self.opt._sync_hyper(1.0 / self.world)
        if self._optst is None:
            dev = self.opt.arena.flat_g.device
            self._comm, self._optst = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        self._optst.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._optst):
            ops.adamw_prep_dev(self.opt._hyper)

    def _end_updates(self):
        torch.cuda.current_stream().wait_stream(self._optst)   # the next forward reads the updated weights (and their bf16 shadows)
        self.opt.step_count += 1
        self.opt._hyper_step = self.opt.step_count
        if self.opt.arena.flat_s is not None:
            self.opt.arena.adopt_shadow()
        else:
            Fn.SHADOWS.invalidate()

    def _run_pieces(self, data, graphs=None):
        g = 1 if self.gather is not None else 0
        n = len(self.cuts) + 1 + g
        if self.collective and self.opt.arena.flat_g.is_cuda:
            self._begin_updates()
        try:
            for k in range(n):
                if graphs is not None:
                    graphs[k].replay()
                else:
                    self._piece(k, data)
                if g and k == 0:
                    self._gather()                       # the embedding all-gather: outside the graphs, between the forward and the loss
                    continue
                if self.collective:
                    self._exchange_and_update(self.piece_runs[k - g], piece=k - g)
        except BaseException:
            # _begin_updates() advanced the device step counter and earlier pieces may already be updated (their moments too): the
            # host count did not move, so have the next _sync_hyper() rewrite the device counter from it
            self.opt._hyper_step = -1
            if self._optst is not None:
                # ... but only once the aborted step's update launches have drained: they still read the hyper-parameter block on the
                # optimizer stream (behind collectives on the communication stream), and the rewrite happens on the current one
                cur = torch.cuda.current_stream()
                cur.wait_stream(self._comm)
                cur.wait_stream(self._optst)
            raise
        if self.collective:
            if self.opt.arena.flat_g.is_cuda:
                self._end_updates()
            else:
                self.opt.launch(grad_scale=1.0 / self.world)

    def _eager(self, data):
        self._run_pieces(data)
        return self.out

    @property
    def inputs(self):
        """The input buffers ``{'text': {input_ids, attention_mask}, 'object', 'object_mask'}`` of the captured set used last (None
        before the first capture; a batch of another shape has its own set, with its own buffers, once it has been captured).
        A loader that writes a batch straight into them -- ``RegionBatcher.to_device(out=step.inputs)``: the selection kernel's
        output IS the model's input -- and hands the same dict to ``__call__`` saves the device-to-device copy of the batch (151 MB
        of region features at B = 64: ~65 us a step).  Write into them only from the thread / stream that replays the graphs:
        stream order is what keeps batch n + 1 from landing before step n has read batch n."""
        return self.static if self.graphs is not None else None

    def inputs_for(self, data):
        """The captured input buffers for a batch of ``data``'s shape, or None while that shape has no capture yet (its first ``warmup``
        calls run eagerly on the caller's own tensors) -- what a loader should ask before ``RegionBatcher.to_device(out=...)``: ``inputs``
        alone keeps naming the LAST replayed set, whose buffers have another shape when the loader's final batch is smaller.
        Every captured shape pins its own activation pool (a step's saved activations: ~9 GB at B = 64), up to ``max_shapes`` of them."""
        have = self._sets.get(self._key(data))
        return have[1] if have is not None else None

    @staticmethod
    def _put(dst, src):
        if src.data_ptr() != dst.data_ptr():                  # (already in place: the caller staged into `inputs`)
            dst.copy_(src, non_blocking=True)

    def _capture(self, data):
        self.static = {"text": {k: v.clone() for k, v in data["text"].items()}, "object": data["object"].clone(),
                       "object_mask": data["object_mask"].clone()}
        self.shape_key = self._key(data)
        torch.cuda.synchronize()
        self.opt._sync_hyper(1.0 / self.world)              # no host->device copy may happen inside the capture
        self.graphs = []
        for k in range(len(self.cuts) + 1 + (1 if self.gather is not None else 0)):     # graphs sharing one memory pool: each consumes what the previous ones saved
            g = torch.cuda.CUDAGraph()
            # ops.graph_capture: thread_local error mode (a loader thread keeps working while this thread captures) and the eager stream's
            # cached scratch (nothing that is cached may live in this graph's private pool)
            with ops.graph_capture(g, pool=self.graphs[0].pool() if self.graphs else None):
                self._piece(k, self.static)
            self.graphs.append(g)
            if self.gather is not None and k == 0:
                self._gather_buffers()                        # (shapes only: the collective itself runs between REPLAYS, never here)
        if not self.collective:
            self.opt.step_count -= 1                          # launch() counted a step, but capturing executed nothing

    def __call__(self, data):
        self.calls += 1
        key = self._key(data)
        have = self._sets.get(key)
        if have is None:
            n = self._seen.get(key, 0) + 1
            self._seen[key] = n
            if n <= self.warmup:
                # a shape's first calls run eagerly: kernel attributes, workspaces and every other first-use side effect that may not
                # happen inside a capture are behind it when the capture comes
                return self._eager(data)
            if len(self._sets) >= self.max_shapes:
                self._sets.pop(next(iter(self._sets)))        # least recently used capture (its memory pool goes with it)
            self._capture(data)
            self._sets[key] = (self.graphs, self.static, self.out, self._fwd, self._gbuf)
        else:
            self._sets[key] = self._sets.pop(key)             # most recently used last: eviction takes the front
            self.graphs, self.static, self.out, self._fwd, self._gbuf = have
            self.shape_key = key
            for k, v in data["text"].items():
                self._put(self.static["text"][k], v)
            self._put(self.static["object"], data["object"])
            self._put(self.static["object_mask"], data["object_mask"])
        if not self.collective:
            # the captured AdamW kernels read lr / betas / eps / wd / the step counter from the device buffer: follow any change the
            # host made since the last call (param_groups[0]['lr'] = ..., load_state_dict) -- outside the graph, copies only on change
            self.opt._sync_hyper(1.0)
            self.graphs[0].replay()
            self.opt.replayed()
        else:
            self._run_pieces(None, self.graphs)
        # fresh tensors, like the eager step (a later replay must not rewrite them): one launch for the three scalars, not three copies
        return tuple(torch.stack([t.reshape(()) for t in self.out]).unbind(0))


def evaluate(model, loss_fn, batches, metrics=None, use_local=True, mscoco=False, log=None, precision=None):
    """Retrieval evaluation of one validation loader, mirroring ``Multi_ObjectTrainer_dist._valid_epoch``
    (trainer/trainer_dist.py:205-408) on already-tokenised batches ``{'text': {input_ids, attention_mask}, 'object', 'object_mask'}``:

      per batch (:236-346)   forward without gradients, all-gather of lengths / masks / the four embedding tensors / the object
                             mask when a process group with more than one rank is up (:252-321), per-batch validation loss
      after the loop (:358-399)  ``o2t_sims = sim_matrix(text, object)`` (text x video) PLUS ``get_sim_by_segment(local_object,
                             local_text, ...)`` (video x text) -- the two addends have transposed orientations and are added
                             element-wise exactly as the reference does (it only type-checks on square eval sets); then every
                             metric on ``o2t_sims``.

    ``precision`` is handed to ``get_sim_by_segment``: ``None`` (default) follows the model -- an fp32 model evaluates its grid on
    the fp32 parity path, a bf16 model on the fused per-pair MFMA kernel; ``'float32'`` forces the parity path for a bf16 model.
    The embeddings stay in HBM (the reference parks them on the host between batches; 1000 MSRVTT pairs are 0.35 GB here).
    Returns ``{'val_loss', 'o2t_sims' (numpy [N,N]), 'global_sims', 'local_sims', 'nested_val_metrics': {name: dict}}``."""
    from . import metric as M
    metrics = metrics if metrics is not None else (M.t2v_metrics, M.v2t_metrics)
    world = dist.get_world_size() if dist.is_initialized() else 1
    gather = (lambda t: _gather_plain(t, world)) if world > 1 else (lambda t: t)
    acc = {k: [] for k in ("gt", "go", "lt", "lo", "len", "om", "tm")}
    total, nb = 0.0, 0
    was_training = model.training
    model.eval()
    with torch.no_grad():
        for data in batches:
            att = data["text"]["attention_mask"]
            text_length = gather(torch.sum(att, dim=1))
            text_mask = (gather(att[:, 1:].contiguous()) - 1.0) * 100.0
            out = model(data, return_embeds=True)
            gt, go = gather(out["global_text_embeddings"]), gather(out["global_object_embeddings"])
            lt, lo = gather(out["local_text_embeddings"]), gather(out["local_object_embeddings"])
            om = gather(out["object_mask"])
            for k, v in zip(acc, (gt, go, lt, lo, text_length, om, text_mask)):
                acc[k].append(v)
            loss, gl, ll = loss_fn(sim_matrix(gt, go), lo, lt, om, text_length, text_mask)
            if log is not None:
                log("loss:{}, global_loss: {}, local_loss: {}".format(loss.item(), gl.item(), ll.item()))
            total += float(loss.item())
            nb += 1
        cat = {k: torch.cat(v) for k, v in acc.items()}
        if mscoco:                                              # :363-366
            cat["go"], cat["lo"], cat["om"] = cat["go"][::5], cat["lo"][::5], cat["om"][::5]
        global_sims = sim_matrix(cat["gt"], cat["go"]).detach().float().cpu().numpy()
        o2t_sims, local_sims = global_sims, None
        if use_local:
            local_sims = loss_fn.local_loss.get_sim_by_segment(cat["lo"], cat["lt"], cat["om"], cat["len"], cat["tm"], device=cat["lo"].device,
                                                               precision=precision)
            o2t_sims = global_sims + local_sims                 # [n_text, n_video] + [n_video, n_text]: the reference's addend quirk
    if was_training:
        model.train()
    nested = {}
    for fn in metrics:
        nested[fn.__name__] = fn(o2t_sims, fold=5) if mscoco else fn(o2t_sims)
    return dict(val_loss=total / max(nb, 1), o2t_sims=o2t_sims, global_sims=global_sims, local_sims=local_sims, nested_val_metrics=nested)
