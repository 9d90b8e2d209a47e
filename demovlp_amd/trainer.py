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
    """HF AdamW (eps added to sqrt(v) before bias correction, decoupled decay, correct_bias=True) over a ParamArena."""

    def __init__(self, arena: ParamArena, lr=1e-5, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0):
        self.arena = arena
        self.param_groups = [dict(params=arena.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.m = torch.zeros_like(arena.flat_p)
        self.v = torch.zeros_like(arena.flat_p)
        self.step_count = 0
        if arena.flat_p.is_cuda:
            ops.enable_deferred_reductions(arena.flat_p.device)

    def zero_grad(self, set_to_none=True):
        for p in self.arena.params:
            p.grad = None
        # gradients of tensors that received none this step must read as 0 for the flat update
        self._need_zero = True

    def step(self, grad_scale=1.0):
        Fn.join_side_stream()          # weight gradients may still be in flight on the side stream
        if self.arena.flat_p.is_cuda:
            ops.flush_reductions()     # bias / LayerNorm gradients: one batched final reduction (no-op when already flushed)
        g = self.param_groups[0]
        a = self.arena
        # tensors without a gradient this step (norm3.*, object_model.norm.*, ...) must not be updated: their slices read zero
        a.zero_untouched(lambda i: a.params[i].grad is not None)
        self.step_count += 1
        ops.adamw_step(a.flat_p, a.flat_g, self.m, self.v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                       self.step_count, grad_scale, a.flat_s)
        if a.flat_s is not None:
            a.adopt_shadow()

    def state_dict(self):
        return dict(step=self.step_count, m=self.m, v=self.v, param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])


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


def train_step(model, loss_fn, optimizer, data, reducer: GradReducer | None = None):
    """One optimisation step on an already-tokenised, already-on-device batch (trainer/trainer_dist.py:144-171)."""
    text_length = torch.sum(data["text"]["attention_mask"], dim=1)
    optimizer.zero_grad()
    if reducer is not None:
        reducer.begin()
    out = model(data)
    text_mask = data["text"]["attention_mask"][:, 1:].contiguous()
    text_mask = (text_mask - 1.0) * 100.0
    global_sim = sim_matrix(out["global_text_embeddings"], out["global_object_embeddings"])
    loss, global_loss, local_loss = loss_fn(global_sim, out["local_object_embeddings"], out["local_text_embeddings"],
                                            out["object_mask"], text_length, text_mask)
    loss.backward()
    if loss.is_cuda:
        Fn.join_side_stream()          # deferred partial sums may have been produced on the side stream
        ops.flush_reductions()
    scale = reducer.finish() if reducer is not None else 1.0
    if isinstance(optimizer, FusedAdamW):
        optimizer.step(grad_scale=scale)
    else:
        if scale != 1.0:
            for p in model.parameters():
                if p.grad is not None:
                    p.grad.mul_(scale)
        optimizer.step()
    return loss.detach(), global_loss.detach(), local_loss.detach()
