"""Round-3 parity coverage on a real MI355X: the code the first multi-GPU run executes, checked numerically with two ranks on one
GPU (gloo carries the collectives, the HIP kernels do everything else):

* the two-graph data-parallel step (``GraphedTrainStep``: backward cut at object block 6, early gradient runs exchanged on a
  communication stream while the second graph runs, late runs behind it, eager optimizer launch) and the one-graph form,
  against a single process that averages the two ranks' gradients by hand (base/base_trainer.py:29-33 semantics);
* ``AllGather_multi`` / ``train_step(gather_negatives=...)`` on the device (trainer/trainer_dist.py:13-31): the 2B x 2B
  contrastive losses against the CPU oracle, backward = the local slice;
* graph replay following a changed learning rate / a resumed step counter;
* the bf16 path at the benchmark size with the measured deviation printed and bounded, gradient norms included, and the bf16
  10-step loss curve through graph replay against the fp64 curve G8b.
"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from demovlp_amd import ops, synthetic as syn  # noqa: E402
from demovlp_amd.loss import GlobalLocalLoss  # noqa: E402
from demovlp_amd.model import ObjectRelation, sim_matrix  # noqa: E402
from demovlp_amd.trainer import (FusedAdamW, GraphedTrainStep, ParamArena, forward_backward, train_step)  # noqa: E402
from helpers import golden_batch, load_golden  # noqa: E402
from oracle import restatement as orc  # noqa: E402

DEV = "cuda"


def build(F, R, dtype="float32"):
    m = ObjectRelation({"model": "", "input_objects": False, "object_num": R, "num_frames": F, "time_module": None},
                       {"model": "pretrained/distilbert-base-uncased", "pretrained": True, "input": "text", "two_outputs": True}, pretrained_init=False,
                       compute_dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(F, R).items()}, strict=True)
    m.set_text_dropout(0.0)
    return m.to(DEV)


def loss_head():
    return GlobalLocalLoss(use_local=True, use_global=True, coef=1.0, focal_type="equal")


def to_dev(obj, mask, ids, att):
    return {"text": {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(att).to(DEV)},
            "object": torch.from_numpy(obj).to(DEV), "object_mask": torch.from_numpy(mask).to(DEV)}


def _np_batch(F, R, B, rank, step):
    obj, mask = syn.fast_region_batch(B, F, R, seed=101 + 7 * step + rank)
    ids, att = syn.caption_batch(B, first_sample=(2 * step + rank) * B)
    return obj, mask, ids, att


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _spawn(target, args_of_rank, world=2, timeout=900):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args_of_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            r = q.get(timeout=timeout)
            if isinstance(r[1], str):             # a rank failed: its peers are waiting in a collective that will never complete -- do not wait for them
                raise AssertionError("rank %d failed:\n%s" % (r[0], r[1]))
            res.append(r)
    except BaseException:
        for p in procs:
            if p.is_alive():
                p.terminate()
        raise
    for p in procs:
        p.join(120)
    return sorted(res, key=lambda t: t[0])


# ---------------------------------------------------------------------------------------------------------------------
# (a) GraphedTrainStep at world 2
# ---------------------------------------------------------------------------------------------------------------------
NSTEP = 5           # 2 eager warm-ups, the capturing call, 2 replays -- a different batch every step


def _graph_dp_worker(rank, world, port, q, cut):
    import traceback
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grad_dtype = "float32"
        if isinstance(cut, tuple) and cut and cut[-1] == "side-stream":
            # weight gradients on the gradient side stream: every piece must join it before its ranges are exchanged
            from demovlp_amd import functional as Fn
            Fn.OVERLAP_WGRAD = 2
            cut = cut[:-1]
        if isinstance(cut, tuple) and cut and cut[-1] == "bf16-buckets":
            grad_dtype, cut = "bfloat16", cut[:-1]              # every bucket crosses as bf16 (cast kernel, bf16 sum, cast back into the fp32 arena)
        F, R, B = 8, 36, 2
        model = build(F, R)
        arena = ParamArena(model)
        opt = FusedAdamW(arena, lr=1e-3)
        stepper = GraphedTrainStep(model, loss_head(), opt, warmup=2, cut=cut, bucket_mb=64.0, grad_dtype=grad_dtype)
        losses = []
        for s in range(NSTEP):
            out = stepper(to_dev(*_np_batch(F, R, B, rank, s)))
            losses.append(float(out[0].item()))
        torch.cuda.synchronize()
        info = dict(graph2=stepper.graph2 is not None, early=len(stepper.early_runs), late=len(stepper.late_runs), steps=opt.step_count,
                    captured=stepper.graph is not None, graphs=len(stepper.graphs or ()))
        q.put((rank, losses, arena.flat_p.double().cpu().numpy()[::211], arena.flat_p.sum().item(), info))
    except BaseException:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


_HAND_REF = []


def _hand_averaged_reference():
    """One process that runs both ranks' batches, sums the gradients itself and steps with grad_scale 1/2 (the same for every variant of the test
    below: computed once)."""
    if _HAND_REF:
        return _HAND_REF[0]
    F, R, B = 8, 36, 2
    model = build(F, R)
    arena = ParamArena(model)
    opt = FusedAdamW(arena, lr=1e-3)
    lf = loss_head()
    ref_losses = []
    for s in range(NSTEP):
        acc, ls = torch.zeros_like(arena.flat_g), []
        for rank in range(2):
            opt.zero_grad()
            loss, _, _ = forward_backward(model, lf, to_dev(*_np_batch(F, R, B, rank, s)))
            ops.flush_reductions()
            arena.zero_untouched(lambda i: arena.params[i].grad is not None)
            acc += arena.flat_g
            ls.append(float(loss.item()))
        ref_losses.append(ls)
        arena.flat_g.copy_(acc)
        opt.step(grad_scale=0.5)
    _HAND_REF.append((ref_losses, arena.flat_p.double().cpu().numpy()[::211]))
    return _HAND_REF[0]


# (default run: the bench's configuration and its bf16-bucket variant; the gradient-side-stream mode (off in bench.py), two graphs and one graph with --runslow)
@pytest.mark.parametrize("cut", [(8, 4), pytest.param((8, 4, "side-stream"), marks=pytest.mark.slow), (8, 4, "bf16-buckets"), pytest.param(6, marks=pytest.mark.slow),
                                 pytest.param(None, marks=pytest.mark.slow)],
                         ids=["three-graphs-cuts8-4", "three-graphs-wgrad-side-stream", "three-graphs-bf16-buckets", "two-graphs-cut6", "one-graph"])
def test_two_rank_graphed_step_matches_hand_averaged_gradients(cut):
    """Two processes on cuda:0, each with its own batch per step, five steps through GraphedTrainStep (two of them replays of the
    captured graphs with the bucketed gradient exchange between / behind them and the fused optimizer applied bucket by bucket on
    its own stream as each reduction completes).  The ranks' parameters must be bit-equal, and equal -- to
    1e-5 of max|p| -- to one process that runs both batches, sums the gradients itself and steps with grad_scale 1/2."""
    res = _spawn(_graph_dp_worker, (cut,))
    (_, l0, p0, s0, info), (_, l1, p1, s1, _) = res
    half = isinstance(cut, tuple) and cut[-1] == "bf16-buckets"
    if isinstance(cut, tuple) and cut[-1] in ("side-stream", "bf16-buckets"):
        cut = cut[:-1]
    assert info["captured"] and info["steps"] == NSTEP
    assert info["graph2"] == (cut is not None) and (info["early"] >= 2 if cut is not None else info["early"] == 0)
    assert info["graphs"] == (1 if cut is None else 2 if isinstance(cut, int) else len(cut) + 1)
    assert np.array_equal(p0, p1) and s0 == s1                       # lock step, bit for bit
    ref_losses, pref = _hand_averaged_reference()
    for s in range(NSTEP):
        for r, l in ((0, l0), (1, l1)):
            assert abs(l[s] - ref_losses[s][r]) < (5e-3 if half else 1e-5) * max(1.0, abs(l[s])), (s, r, l[s], ref_losses[s][r])
    dev = np.abs(p0 - pref).max() / max(1.0, np.abs(pref).max())
    print("\nparameters after %d data-parallel graph steps vs hand-averaged reference: max rel dev %.3g" % (NSTEP, dev))
    # bf16 buckets: the summed gradients carry three bf16 roundings (2^-9 each); Adam's m / sqrt(v) turns that into <= a few per cent of lr = 1e-3
    # per step on some elements -- observed 1.6e-3 after five steps; the fp32 exchange is exact to the hand-averaged reference
    assert dev <= (5e-3 if half else 1e-5)


# ---------------------------------------------------------------------------------------------------------------------
# (b) AllGather_multi / gather_negatives on the device
# ---------------------------------------------------------------------------------------------------------------------
GRAD_KEYS = ("txt_proj.1.weight", "object_model.proj.weight", "object_model.blocks.11.mlp.fc2.weight", "object_model.blocks.0.attn.qkv.weight",
             "text_model.transformer.layer.5.ffn.lin2.weight", "object_model.cls_token", "object_model.temporal_embed")


def _gather_worker(rank, world, port, q, with_step):
    import argparse
    import traceback
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        F, R, B = 8, 36, 2
        model = build(F, R)
        arena = ParamArena(model)
        opt = FusedAdamW(arena, lr=1e-3)
        args = argparse.Namespace(world_size=world, rank=rank)
        data = to_dev(*_np_batch(F, R, B, rank, 0))
        opt.zero_grad()
        loss, gl, ll = forward_backward(model, loss_head(), data, gather_negatives=args)
        ops.flush_reductions()
        torch.cuda.synchronize()
        named = dict(model.named_parameters())
        grads = {k: named[k].grad.detach().double().cpu().numpy() for k in GRAD_KEYS}
        step_losses = None
        if with_step:            # the whole optimisation step with the hook-driven reducer (what bench.py --gather-negatives --graph 0 runs)
            from demovlp_amd.trainer import GradReducer
            red = GradReducer(arena, bucket_mb=64.0)
            step_losses = [float(train_step(model, loss_head(), opt, data, red, gather_negatives=args)[0].item()) for _ in range(2)]
            torch.cuda.synchronize()
        q.put((rank, [loss.item(), gl.item(), ll.item()], grads, step_losses, arena.flat_p.double().cpu().numpy()[::211]))
    except BaseException:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _graph_gather_worker(rank, world, port, q):
    """GraphedTrainStep(gather_negatives=args): forward graph -> all-gather outside the graphs -> loss / backward graphs + bucketed gradient exchange."""
    import argparse
    import faulthandler
    import sys
    import traceback
    import torch.distributed as dist
    faulthandler.dump_traceback_later(150, exit=True, file=sys.stderr)        # a rank stuck in a collective must say where (and end) instead of timing the test out
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        F, R, B = 8, 36, 2
        model = build(F, R)
        arena = ParamArena(model)
        opt = FusedAdamW(arena, lr=1e-3)
        args = argparse.Namespace(world_size=world, rank=rank)
        data = to_dev(*_np_batch(F, R, B, rank, 0))
        stepper = GraphedTrainStep(model, loss_head(), opt, warmup=2, gather_negatives=args)
        losses, after2 = [], None
        dbg = os.environ.get("DVLP_GATHER_DEBUG")

        def mark(msg):
            if dbg:
                with open(os.path.join(dbg, "gather_rank%d.log" % rank), "a") as fh:
                    fh.write(msg + "\n")
        if dbg:                      # where does a rank stop?  (developer aid: DVLP_GATHER_DEBUG=<dir>)
            import demovlp_amd.trainer as T
            for name in ("_piece", "_gather", "_exchange_and_update", "_capture", "_begin_updates", "_end_updates"):
                orig = getattr(T.GraphedTrainStep, name)

                def wrap(self_, *a, _o=orig, _n=name, **k):
                    mark("enter %s %s" % (_n, a[0] if a and isinstance(a[0], int) else ""))
                    r = _o(self_, *a, **k)
                    mark("leave %s" % _n)
                    return r
                setattr(T.GraphedTrainStep, name, wrap)
        for s in range(5):                                   # 2 eager warm-ups, the capturing call, 2 replays
            mark("step %d" % s)
            losses.append([float(t.item()) for t in stepper(data)])
            if s == 1:
                torch.cuda.synchronize()
                after2 = arena.flat_p.double().cpu().numpy()[::211]
        # a batch of ANOTHER shape (a loader's smaller last batch): its own warm-up, capture, buffers; then the first shape again
        small = to_dev(*_np_batch(F, R, 1, rank, 3))
        for s in range(4):
            losses.append([float(t.item()) for t in stepper(small)])
        losses.append([float(t.item()) for t in stepper(data)])
        torch.cuda.synchronize()
        info = dict(graphs=len(stepper.graphs or ()), sets=len(stepper._sets), steps=opt.step_count, gbufs=len(stepper._gbufs))
        faulthandler.cancel_dump_traceback_later()
        q.put((rank, losses, after2, arena.flat_p.double().cpu().numpy()[::211], info))
    except BaseException:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gather_negatives_two_ranks_on_device_vs_oracle():
    """AllGather_multi on device tensors at world 2: every rank computes GlobalLocalLoss over the 2B = 4 pairs of both ranks
    (forward within 1e-4 of the oracle's 4 x 4 loss, identical on both ranks), and its backward carries only the LOCAL slice of the
    gathered tensors' gradient (trainer_dist.py:25-31): parameter gradients equal the oracle's with the other rank's embeddings
    detached.  Then two full train steps with the gradient all-reduce: ranks stay bit-equal."""
    res = _spawn(_gather_worker, (True,))
    F, R, B = 8, 36, 2
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    for rank in range(2):
        p = orc.params_from_numpy(syn.fill_state_dict(F, R), requires_grad=True)
        outs, tms = [], []
        for r in range(2):
            obj, mask, ids, att = _np_batch(F, R, B, r, 0)
            ctx = torch.enable_grad() if r == rank else torch.no_grad()
            with ctx:
                outs.append(orc.model_forward(p, torch.from_numpy(ids), torch.from_numpy(att), torch.from_numpy(obj), torch.from_numpy(mask)))
            tms.append((torch.from_numpy(att)[:, 1:].float() - 1.0) * 100.0)
        out = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
        loss, g, l, _, _ = orc.global_local_loss(out, torch.cat(tms, 0))
        loss.backward()
        _, got, grads, step_losses, _ = res[rank]
        ref = np.array([loss.item(), g.item(), l.item()])
        assert np.abs(np.array(got) - ref).max() < 1e-4 * max(1.0, abs(ref[0])), (rank, got, ref)
        for k in GRAD_KEYS:
            rg = p[k].grad.double().numpy()
            err = np.abs(grads[k] - rg).max() / max(1e-12, np.abs(rg).max())
            assert err < 2e-3, (rank, k, err)                # fp32 MFMA vs CPU fp32 accumulation order; the e2e goldens hold the same bar
        assert step_losses is not None and abs(step_losses[0] - ref[0]) < 1e-4 * max(1.0, abs(ref[0]))
        assert np.isfinite(step_losses[1]) and step_losses[1] != step_losses[0]          # the update happened (lr 1e-3 overshoots on 4 pairs)
    assert np.abs(np.array(res[0][1]) - np.array(res[1][1])).max() < 1e-6 * abs(res[0][1][0])   # same 4 x 4 loss on both ranks
    assert np.array_equal(res[0][4], res[1][4])
    # The same through GraphedTrainStep(gather_negatives=...) (round 6: bench.py --gather-negatives no longer drops to eager mode): the
    # forward graph ends at the local embeddings, the all-gather runs between two replays, the loss / backward graphs start from the gathered
    # buffers, the three-piece gradient exchange follows.  Its first two steps (eager warm-ups) and its replays must reproduce the eager
    # train_step sequence above; ranks bit-equal; a batch of another shape gets buffers of its own.
    gres = _spawn(_graph_gather_worker, (), timeout=240)
    for rank in range(2):
        _, gl, after2, pfin, info = gres[rank]
        eager = res[rank][3]
        assert abs(gl[0][0] - eager[0]) < 1e-5 * abs(eager[0]) and abs(gl[1][0] - eager[1]) < 1e-5 * abs(eager[1]), (gl[:2], eager)
        assert np.abs(after2 - res[rank][4]).max() <= 1e-5 * max(1.0, np.abs(res[rank][4]).max())          # parameters after two steps = the eager path's
        assert info["graphs"] == 4 and info["sets"] == 2 and info["gbufs"] == 2 and info["steps"] == 10, info
        assert all(np.isfinite(x).all() for x in gl)
    assert np.array_equal(gres[0][2], gres[1][2]) and np.array_equal(gres[0][3], gres[1][3])             # lock step through capture and replay
    assert np.abs(np.array(gres[0][1]) - np.array(gres[1][1])).max() < 1e-6 * abs(gres[0][1][0][0])       # every rank sees the same 2B x 2B loss
    # replays continue the eager sequence: the loss keeps moving and the 5th call (2nd replay) differs from the 3rd (the capturing call ran nothing,
    # the replay behind it did)
    assert gres[0][1][2] != gres[0][1][3] != gres[0][1][4]


# ---------------------------------------------------------------------------------------------------------------------
# a replayed graph must follow lr changes and a resumed step counter
# ---------------------------------------------------------------------------------------------------------------------
def test_graph_replay_follows_lr_change_and_resumed_step_counter(tmp_path):
    from demovlp_amd.trainer import resume_checkpoint, save_checkpoint
    F, R, B = 8, 36, 2
    data = to_dev(*golden_batch(F, R, B))
    finals = []
    for graphed in (False, True):
        from demovlp_amd import functional as Fn
        Fn.SHADOWS.clear()
        model = build(F, R)
        arena = ParamArena(model)
        opt = FusedAdamW(arena, lr=1e-3)
        lf = loss_head()
        stepper = GraphedTrainStep(model, lf, opt, warmup=2) if graphed else None
        run = (lambda: stepper(data)) if graphed else (lambda: train_step(model, lf, opt, data))
        for _ in range(4):
            run()                                              # graphed: 2 eager, capture, 1 replay
        opt.param_groups[0]["lr"] = 0.0                        # the reference's _adjust_learning_rate writes param_groups the same way
        before = arena.flat_p.clone()
        run()
        torch.cuda.synchronize()
        assert torch.equal(before, arena.flat_p), "lr = 0 after capture must freeze the parameters (graphed=%s)" % graphed
        opt.param_groups[0]["lr"] = 5e-4
        run()
        # resume: the step counter (bias correction) comes from the checkpoint, the captured kernels must pick it up
        ck = str(tmp_path / ("ck%d.pth" % graphed))
        save_checkpoint(ck, model, opt, epoch=1)
        sd = torch.load(ck, map_location="cpu", weights_only=False)
        for st in sd["optimizer"]["state"].values():
            st["step"] = 40
        torch.save(sd, ck)
        resume_checkpoint(ck, model, opt)
        assert opt.step_count == 40
        run()
        torch.cuda.synchronize()
        assert opt.step_count == 41
        finals.append(arena.flat_p.clone())
    assert torch.equal(finals[0], finals[1])


# ---------------------------------------------------------------------------------------------------------------------
# (c) bf16 at the benchmark size: measured deviation, bounded at 2x what was observed; gradient norms
# ---------------------------------------------------------------------------------------------------------------------
BF16_B64_LOSS_TOL = 1.5e-4        # relative to the total loss: ~2x the observed 6.2e-5 (max abs deviation 0.0011 on 17.83; printed)
BF16_B64_GRADNORM_TOL = 1.5e-2    # relative, per tensor: 2x the observed maximum (7.5e-3, object_model.blocks.11.attn.qkv.bias; printed)


def test_bf16_at_benchmark_size_losses_and_gradient_norms():
    """B=64, F=8, R=36 (what bench.py times).  Losses of the bf16 step against the CPU oracle; gradient norms of
    object_model.blocks.{0,11}.* and txt_proj.1.weight against the fp32 HIP path on the same batch (that path is held to 1e-4 /
    the reference's gradient norms by the goldens; the oracle's B=64 backward would need ~10 GB of host memory)."""
    F, R, B = 8, 36, 64
    obj, mask = syn.fast_region_batch(B, F, R, seed=7)
    ids, att = syn.caption_batch(B)
    data = to_dev(obj, mask, ids, att)
    res = {}
    for dtype in ("float32", "bfloat16"):
        from demovlp_amd import functional as Fn
        Fn.SHADOWS.clear()
        model = build(F, R, dtype)
        arena = ParamArena(model, bf16_shadow=(dtype == "bfloat16"))
        opt = FusedAdamW(arena, lr=1e-5)
        opt.zero_grad()
        losses = forward_backward(model, loss_head(), data)
        ops.flush_reductions()
        torch.cuda.synchronize()
        norms = {n: float(p.grad.double().norm()) for n, p in model.named_parameters()
                 if p.grad is not None and (n.startswith(("object_model.blocks.0.", "object_model.blocks.11.")) or n == "txt_proj.1.weight")}
        res[dtype] = (np.array([float(x.item()) for x in losses]), norms)
        del model, arena, opt
        torch.cuda.empty_cache()
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    p = orc.params_from_numpy(syn.fill_state_dict(F, R))
    with torch.no_grad():
        out = orc.model_forward(p, torch.from_numpy(ids), torch.from_numpy(att), torch.from_numpy(obj), torch.from_numpy(mask))
        tm = (torch.from_numpy(att)[:, 1:].float() - 1.0) * 100.0
        ref = np.array([x.item() for x in orc.global_local_loss(out, tm, batched=True)[:3]])
    d32 = np.abs(res["float32"][0] - ref).max() / ref[0]
    d16 = np.abs(res["bfloat16"][0] - ref).max() / ref[0]
    n32, n16 = res["float32"][1], res["bfloat16"][1]
    assert len(n32) >= 20 and set(n32) == set(n16)
    gdev = {k: abs(n16[k] - n32[k]) / n32[k] for k in n32 if n32[k] > 1e-6}
    worst = max(gdev, key=gdev.get)
    print("\nB=64 losses: oracle %s | fp32 HIP rel dev %.2e | bf16 HIP %s max_abs_dev_vs_oracle %.4f (rel %.2e)" %
          (np.array2string(ref, precision=4), d32, np.array2string(res["bfloat16"][0], precision=4), np.abs(res["bfloat16"][0] - ref).max(), d16))
    print("B=64 bf16 gradient norms vs fp32 HIP over %d tensors: max rel dev %.3e (%s)" % (len(gdev), gdev[worst], worst))
    assert d32 < 1e-4
    assert d16 < BF16_B64_LOSS_TOL, (res["bfloat16"][0], ref)
    assert gdev[worst] < BF16_B64_GRADNORM_TOL, (worst, gdev[worst])


# ---------------------------------------------------------------------------------------------------------------------
# (d) bf16 10-step loss curve through graph replay against the fp64 curve
# ---------------------------------------------------------------------------------------------------------------------
BF16_CURVE_HEAD_TOL = 4e-2        # steps 1-4, relative to the step's total loss: 2x the largest deviation observed on MI355X (6e-3 at lr 1e-5, 2.1e-2 at lr 2e-4; printed)
BF16_CURVE_TAIL_GAP = 1.0         # steps 5-10: the bf16 loss may trail the fp64 loss by at most this much (observed <= 0.89), and must keep falling


@pytest.mark.parametrize("tag,lr", [("lr1e-5", 1e-5), ("lr2e-4", 2e-4)])
def test_bf16_ten_step_loss_curve_through_graph_replay_vs_f64_curve(tag, lr):
    """Golden G8b (the 10-step curve in float64, tests/golden/make_f64_curve.py) with the bf16 model, optimizer-written bf16 shadow
    weights and hipGraph replay -- the configuration bench.py measures.

    What bf16 can and cannot track (measured, printed): the first four steps follow the fp64 curve to 6e-3 of the loss (bf16
    activations).  After that the B = 2 toy problem enters the regime where the fp64 run drives the loss from 2.9 to 0.6 with
    updates of lr = 1e-5 per step -- far below the bf16 resolution (2^-9 relative) of most weights, so the bf16 SHADOW weights the
    forward pass reads move later than the fp32 masters: the bf16 run keeps descending (7.36 -> 1.15) but trails the fp64 run by up
    to 0.9.  That is a property of bf16 shadows at this learning rate, not of a kernel; the fp32 path carries the 1e-3 curve bar
    (tests/test_gpu_round2.py).  Asserted: head within BF16_CURVE_HEAD_TOL, tail never more than BF16_CURVE_TAIL_GAP above the fp64
    loss, overall descent."""
    g64 = load_golden("g8b_loss_curve_f64.npz")
    F, R, B = 8, 36, 2
    from demovlp_amd import functional as Fn
    Fn.SHADOWS.clear()
    model = build(F, R, "bfloat16")
    arena = ParamArena(model, bf16_shadow=True)
    opt = FusedAdamW(arena, lr=lr)
    stepper = GraphedTrainStep(model, loss_head(), opt, warmup=2)
    data = to_dev(*golden_batch(F, R, B))
    curve = np.array([[float(x.item()) for x in stepper(data)] for _ in range(10)])
    dev = np.abs(curve - g64[tag]) / np.maximum(1.0, np.abs(g64[tag][:, :1]))
    print("\nG8b", tag, "bf16 graph-replayed curve, relative deviation per step:", np.array2string(dev.max(axis=1), precision=4))
    print("    bf16 total loss:", np.array2string(curve[:, 0], precision=4), "\n    fp64 total loss:", np.array2string(g64[tag][:, 0], precision=4))
    assert stepper.graph is not None
    assert dev[:4].max() < BF16_CURVE_HEAD_TOL, dev.max(axis=1)
    assert (curve[4:, 0] - g64[tag][4:, 0]).max() < BF16_CURVE_TAIL_GAP, (curve[:, 0], g64[tag][:, 0])
    if tag == "lr1e-5":
        assert curve[-1, 0] < 0.35 * curve[0, 0] and curve[6:, 0].mean() < curve[2:6, 0].mean()        # and it trains
    else:       # lr 2e-4 overshoots on this toy batch (the fp64 curve itself goes 7.37 -> 9.95 -> 7.63); bf16 follows it at every step
        assert dev.max() < 0.12, dev.max(axis=1)                                                         # 2x the observed 0.06


# ---------------------------------------------------------------------------------------------------------------------
# 224-row tiles of the 256-row GEMM kernel (round quantisation): the same arithmetic per output element, so bit-equal results
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(18496, 2304, 768), (18496, 768, 3072), (2000, 768, 768), (225, 256, 128), (6400, 3072, 768)])
def test_gemm_short_tiles_bit_equal_to_256_row_tiles(M, N, K):
    """dvlp_dev_gemm_p8_short_tiles: every epilogue kind of the forward / dX products (bias, +residual, GELU with pre-activation out,
    GELU' with pre-activation in and fused column sums) on 224-row tiles against 256-row tiles: identical bits, ragged last row
    tile included (18496 = 82 x 224 + 128; 2000 = 8 x 224 + 208; 225 = 224 + 1)."""
    g = torch.Generator(device=DEV).manual_seed(3)
    bf = torch.bfloat16
    x = torch.randn(M, K, device=DEV, generator=g).to(bf)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(bf)
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g).to(bf)
    dy = torch.randn(M, N, device=DEV, generator=g).to(bf)
    pre = torch.randn(M, K, device=DEV, generator=g).to(bf)
    ops.call("dvlp_dev_gemm_p8_mode", 2)
    ops.call("dvlp_dev_gemm_p8_persistent", 0)          # tile height alone (the persistent form adds the bias first: its own test below)
    ops.enable_deferred_reductions(torch.device(DEV), workspace_mb=64)
    try:
        outs = []
        for mode in (0, 2):
            ops.call("dvlp_dev_gemm_p8_short_tiles", mode)
            aux = torch.empty(M, N, device=DEV, dtype=bf)
            cs = torch.zeros(K, device=DEV)
            o = [ops.linear_fwd(x, w, bias), ops.linear_fwd(x, w, bias, res=res), ops.linear_fwd(x, w, bias, gelu_aux=aux), aux,
                 ops.linear_bwd_input(dy, w), ops.linear_bwd_input(dy, w, gelu_pre=pre, colsum_to=cs)]
            ops.flush_reductions()
            outs.append([t.clone() for t in o] + [cs.clone()])
        for a, b in zip(outs[0][:-1], outs[1][:-1]):
            assert torch.equal(a, b)
        assert torch.allclose(outs[0][-1], outs[1][-1], rtol=1e-5, atol=1e-4)      # partial sums grouped by row tile: fp32 rounding only
        ref = x.float() @ w.float().t() + bias
        assert float((outs[1][0].float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())
    finally:
        ops.disable_deferred_reductions()
        ops.call("dvlp_dev_gemm_p8_short_tiles", 1)
        ops.call("dvlp_dev_gemm_p8_mode", 1)
        ops.call("dvlp_dev_gemm_p8_persistent", 1)


# ---------------------------------------------------------------------------------------------------------------------
# eval path: the grid on the fused per-pair kernel for bf16 models, the precision knob, the MSCOCO branch
# ---------------------------------------------------------------------------------------------------------------------
BF16_EVAL_SIM_TOL = 4e-2          # bf16 o2t / local similarities against the reference's fp32 ones, relative to max|ref|: 2x the observed 1.9e-2 (o2t; local 9.4e-3)


def _eval_batches(F, R, BS, NB):
    from helpers import eval_batch
    return [to_dev(*eval_batch(F, R, BS, b * BS)) for b in range(NB)]


def test_bf16_evaluate_runs_the_grid_on_the_fused_kernel_vs_reference_golden():
    """Golden G9 (the reference's retrieval evaluation, 256 MSRVTT-shape pairs) with a bf16 model: ``evaluate`` hands bf16 embeddings to
    ``get_sim_by_segment``, which then runs the 256 x 256 local grid on the fused per-pair kernel (checked by forcing the multi-kernel
    path and comparing).  Similarities within BF16_EVAL_SIM_TOL of the reference; R@K within what that noise can move (stated)."""
    from demovlp_amd.trainer import evaluate
    g = load_golden("g9_eval.npz")
    F, R, BS, NB = int(g["F"]), int(g["R"]), int(g["batch"]), int(g["batches"])
    n = BS * NB
    model = build(F, R, "bfloat16")
    res = evaluate(model, loss_head(), _eval_batches(F, R, BS, NB))
    ops.call("dvlp_dev_xattn_fused_mode", 0)
    try:
        res_multi = evaluate(model, loss_head(), _eval_batches(F, R, BS, NB))
    finally:
        ops.call("dvlp_dev_xattn_fused_mode", 1)
    res32 = evaluate(model, loss_head(), _eval_batches(F, R, BS, NB), precision="float32")      # bf16 towers, fp32 grid
    scale = np.abs(g["local_sims"]).max()
    d_fused = np.abs(res["local_sims"] - g["local_sims"]).max() / scale
    d_multi = np.abs(res_multi["local_sims"] - g["local_sims"]).max() / scale
    d_f32grid = np.abs(res32["local_sims"] - g["local_sims"]).max() / scale
    d_o2t = np.abs(res["o2t_sims"] - g["o2t_sims"]).max() / np.abs(g["o2t_sims"]).max()
    print("\nbf16 evaluate vs G9 (%d pairs): local sims rel dev fused %.3e / bf16 multi-kernel %.3e / fp32 grid on bf16 embeddings %.3e; o2t %.3e; "
          "fused vs multi-kernel %.3e" % (n, d_fused, d_multi, d_f32grid, d_o2t, np.abs(res["local_sims"] - res_multi["local_sims"]).max() / scale))
    assert not np.array_equal(res["local_sims"], res_multi["local_sims"])           # two different kernels did run
    assert d_fused < BF16_EVAL_SIM_TOL and d_multi < BF16_EVAL_SIM_TOL and d_f32grid < BF16_EVAL_SIM_TOL and d_o2t < BF16_EVAL_SIM_TOL
    assert abs(res["val_loss"] - g["val_losses"][:, 0].mean()) < 3e-2 * g["val_losses"][0, 0]
    keys = ("R1", "R5", "R10", "R50")
    for name in ("t2v", "v2t"):
        got = res["nested_val_metrics"][name + "_metrics"]
        dr = np.abs(np.array([got[k] for k in keys]) - g[name][:4]).max()
        print("   ", name, "R@1/5/10/50 bf16", [round(got[k], 2) for k in keys], "reference", np.round(g[name][:4], 2), "MedR", got["MedR"], g[name][4])
        assert dr <= 100.0 * 8 / n + 1e-9, (name, got, g[name])          # at most 8 of the n queries cross a cut-off under bf16 noise
        assert abs(got["MedR"] - g[name][4]) <= 2.0


def test_get_sim_by_segment_precision_knob():
    from demovlp_amd.loss import RWALoss
    from helpers import eval_grid_inputs
    g = load_golden("g7_metrics.npz")
    im, cap, m_img, lens, m_cap = (torch.from_numpy(x) for x in eval_grid_inputs())
    rwa = RWALoss(20, "equal")
    s32 = rwa.get_sim_by_segment(im, cap, m_img, lens, m_cap, device="cuda")
    s16 = rwa.get_sim_by_segment(im, cap, m_img, lens, m_cap, device="cuda", precision="bfloat16")
    s16b = rwa.get_sim_by_segment(im.bfloat16(), cap.bfloat16(), m_img, lens, m_cap, device="cuda")          # follows the embeddings
    s32b = rwa.get_sim_by_segment(im.bfloat16(), cap.bfloat16(), m_img, lens, m_cap, device="cuda", precision="float32")
    scale = np.abs(g["grid_sims"]).max()
    assert np.abs(s32 - g["grid_sims"]).max() < 1e-4 * scale
    assert np.allclose(s16, s16b, rtol=1e-5, atol=0) and not np.allclose(s16, s32, rtol=1e-5, atol=0)      # (the fused kernel's final sums are not bit-reproducible)
    assert np.abs(s16 - g["grid_sims"]).max() < 2e-2 * scale and np.abs(s32b - g["grid_sims"]).max() < 2e-2 * scale
    with pytest.raises(ValueError):
        rwa.get_sim_by_segment(im, cap, m_img, lens, m_cap, device="cuda", precision="fp8")


def test_evaluate_mscoco_branch_subsamples_videos_and_passes_fold():
    """trainer_dist.py:363-366, 388-389: with a config named MSCOCO* the reference keeps every fifth video (5 captions per image) and
    calls ``metric(o2t_sims, fold=5)``.  Its shipped metrics take no ``fold`` (that call raises TypeError there, and here), so the
    branch is exercised with a metric that does; similarities against the oracle's rectangular sim_matrix, ranks through the
    golden-pinned t2v / v2t metrics with 5 queries per video."""
    from demovlp_amd import metric as M
    from demovlp_amd.trainer import evaluate
    F, R, BS, NB = 8, 30, 10, 2
    model = build(F, R)
    batches = _eval_batches(F, R, BS, NB)
    seen = {}

    def t2v_fold(sims, fold=None):
        seen["fold"], seen["shape"] = fold, sims.shape
        return M.t2v_metrics(sims)

    def v2t_fold(sims, fold=None):
        return M.v2t_metrics(sims)
    res = evaluate(model, loss_head(), batches, metrics=(t2v_fold, v2t_fold), use_local=False, mscoco=True)
    assert seen == {"fold": 5, "shape": (BS * NB, BS * NB // 5)} and res["o2t_sims"].shape == (20, 4) and res["local_sims"] is None
    from helpers import eval_batch
    p = orc.params_from_numpy(syn.fill_state_dict(F, R))
    gt, go = [], []
    with torch.no_grad():
        for b in range(NB):
            obj, mask, ids, att = eval_batch(F, R, BS, b * BS)
            o = orc.model_forward(p, torch.from_numpy(ids), torch.from_numpy(att), torch.from_numpy(obj), torch.from_numpy(mask))
            gt.append(o["global_text_embeddings"]); go.append(o["global_object_embeddings"])
        ref = orc.sim_matrix(torch.cat(gt), torch.cat(go)[::5]).numpy()
    assert np.abs(res["o2t_sims"] - ref).max() < 1e-4
    assert res["nested_val_metrics"]["t2v_fold"] == M.t2v_metrics(res["o2t_sims"]) and res["nested_val_metrics"]["v2t_fold"]["R1"] >= 0.0
    with pytest.raises(TypeError):                              # the reference's own metric functions do not take `fold`
        evaluate(model, loss_head(), batches, use_local=False, mscoco=True)
    with pytest.raises(ValueError):                             # [n_text, n_video] + [n_video, n_text]: the reference's addend only adds up on square sets
        evaluate(model, loss_head(), batches, metrics=(t2v_fold,), use_local=True, mscoco=True)


@pytest.mark.parametrize("M,N,K", [(18496, 2304, 768), (18496, 3072, 256), (9000, 2304, 768), (30000, 768, 2304)])
def test_persistent_gemm_kernel_agrees_with_the_one_tile_form(M, N, K):
    """dvlp_dev_gemm_p8_persistent (default on): one workgroup per CU walks its 224-row tiles, the next tile's first eight units are staged
    across the tile boundary, accumulators start from the bias.  Same products in the same order; the bias enters the fp32 sum first
    instead of last, so outputs with a bias may differ by one bf16 rounding -- bounded here against an fp32 reference; without a
    bias (forward and dX forms) they are bit-equal.  Shapes: 4 / 12 / 36 K tiles, a ragged last row tile (9000 = 40 x 224 + 40), and a
    repeat screen of the cross-tile prefetch (a race would show as run-to-run differences)."""
    g = torch.Generator(device=DEV).manual_seed(5)
    bf = torch.bfloat16
    x = torch.randn(M, K, device=DEV, generator=g).to(bf)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(bf)
    dy = torch.randn(M, N, device=DEV, generator=g).to(bf)
    bias = torch.randn(N, device=DEV, generator=g)
    ref = x.float() @ w.float().t() + bias
    try:
        outs = {}
        for mode in (0, 1):
            ops.call("dvlp_dev_gemm_p8_persistent", mode)
            aux = torch.empty(M, N, device=DEV, dtype=bf)
            outs[mode] = (ops.linear_fwd(x, w, bias), ops.linear_fwd(x, w, bias, gelu_aux=aux), aux, ops.linear_fwd(x, w, None), ops.linear_bwd_input(dy, w))
            if mode == 1:
                for _ in range(12):
                    assert torch.equal(ops.linear_fwd(x, w, bias), outs[1][0]) and torch.equal(ops.linear_bwd_input(dy, w), outs[1][4])
        scale = float(ref.abs().max())
        for a, b in zip(outs[0][:3], outs[1][:3]):
            assert float((a.float() - b.float()).abs().max()) <= 2.0 ** -7 * scale          # within one bf16 rounding of the largest value
        assert torch.equal(outs[0][3], outs[1][3]) and torch.equal(outs[0][4], outs[1][4])  # no bias: bit-equal
        assert float((outs[1][0].float() - ref).abs().max()) < 1e-2 * scale
        assert float((outs[1][2].float() - ref).abs().max()) < 1e-2 * scale                 # pre-activation out of the GELU epilogue
        dref = dy.float() @ w.float()
        assert float((outs[1][4].float() - dref).abs().max()) < 1e-2 * float(dref.abs().max())
    finally:
        ops.call("dvlp_dev_gemm_p8_persistent", 1)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 5 (32-frame long-video variant): bf16 backward, region select at F = 32
# ---------------------------------------------------------------------------------------------------------------------
def test_bf16_32_frames_backward_gradient_norms_vs_fp32_path():
    """F=32, R=36 (N = 1153 tokens per clip, G = 1152 regions in the local loss: the chunked general-G softmax path), bf16 against the fp32 HIP
    path on the same batch (that path is held to the oracle's gradients at 2e-3 by test_fp32_32_frames_backward_vs_oracle_gradients):
    loss within 3e-2, every gradient norm within 5 % (observed maximum printed), none missing."""
    F, R, B = 32, 36, 2
    data = to_dev(*golden_batch(F, R, B))
    res = {}
    for dtype in ("float32", "bfloat16"):
        from demovlp_amd import functional as Fn
        Fn.SHADOWS.clear()
        model = build(F, R, dtype)
        arena = ParamArena(model, bf16_shadow=(dtype == "bfloat16"))
        opt = FusedAdamW(arena, lr=1e-5)
        opt.zero_grad()
        losses = forward_backward(model, loss_head(), data)
        ops.flush_reductions()
        torch.cuda.synchronize()
        res[dtype] = (float(losses[0].item()), {n: float(p.grad.double().norm()) for n, p in model.named_parameters() if p.grad is not None})
        del model, arena, opt
    (l32, n32), (l16, n16) = res["float32"], res["bfloat16"]
    assert set(n32) == set(n16) and len(n32) == 254
    dev = {k: abs(n16[k] - n32[k]) / n32[k] for k in n32 if n32[k] > 1e-5}
    worst = max(dev, key=dev.get)
    print("\nF=32 bf16 vs fp32 HIP: loss %.4f vs %.4f; gradient norms over %d tensors: max rel dev %.3e (%s)" % (l16, l32, len(dev), dev[worst], worst))
    assert abs(l16 - l32) < 3e-2 * abs(l32)
    assert dev[worst] < 5e-2, (worst, dev[worst])


def test_region_select_bit_exact_at_32_frames():
    """K1 at the long-video shape: 32 frames per clip, ragged region counts (28 / 33 / 36 / 50 by sample), R = 36: indices bit-exact, features,
    geometry and mask equal to the oracle's."""
    from helpers import n_raw_for, oracle_clip
    F, R = 32, 36
    for sample in (0, 1, 2, 3):
        nraw = n_raw_for(sample)
        frames = [syn.make_frame(sample, f, nraw) for f in range(F)]
        feats = torch.from_numpy(np.stack([fr["x"] for fr in frames])[None]).to(DEV)
        bbox = torch.from_numpy(np.stack([fr["bbox"] for fr in frames])[None]).to(DEV)
        conf = torch.from_numpy(np.stack([fr["objects_conf"] for fr in frames])[None]).to(DEV)
        wh = torch.tensor([[[640.0, 360.0]] * F], device=DEV)
        obj, mask, order, lens = ops.region_select(feats, bbox, conf, wh, R)
        ref_obj, ref_mask, ref_lens, ref_orders = oracle_clip(sample, F, R)
        assert lens[0].tolist() == ref_lens
        for f in range(F):
            assert order[0, f, : ref_lens[f]].tolist() == ref_orders[f].tolist()
        assert np.array_equal(obj[0].cpu().numpy(), ref_obj) and np.array_equal(mask[0].cpu().numpy(), ref_mask.astype(np.float32))


# ---------------------------------------------------------------------------------------------------------------------
# retrieval evaluation with two ranks (the 2-GPU leg of configs/ft/msrvtt_o2t-select.json): every rank forwards its half of each
# batch, all-gathers embeddings / lengths / masks (trainer_dist.py:252-321) and must arrive at the single-process matrices
# ---------------------------------------------------------------------------------------------------------------------
def _eval_worker(rank, world, port, q):
    import traceback
    import torch.distributed as dist
    from demovlp_amd.trainer import evaluate
    from helpers import eval_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        F, R, BS, NB = 8, 30, 16, 2
        model = build(F, R)
        half = BS // world
        batches = []
        for b in range(NB):
            obj, mask, ids, att = eval_batch(F, R, BS, b * BS)
            sl = slice(rank * half, (rank + 1) * half)
            batches.append(to_dev(obj[sl], mask[sl], ids[sl], att[sl]))
        res = evaluate(model, loss_head(), batches)
        q.put((rank, res["o2t_sims"], res["val_loss"], res["nested_val_metrics"]["t2v_metrics"]["R5"]))
    except BaseException:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_evaluate_with_two_ranks_matches_one_process():
    from demovlp_amd.trainer import evaluate
    res = _spawn(_eval_worker, ())
    F, R, BS, NB = 8, 30, 16, 2
    model = build(F, R)
    one = evaluate(model, loss_head(), _eval_batches(F, R, BS, NB))
    for rank in range(2):
        assert res[rank][1].shape == (BS * NB, BS * NB)
        assert np.abs(res[rank][1] - one["o2t_sims"]).max() < 1e-5 * np.abs(one["o2t_sims"]).max()
        assert abs(res[rank][2] - one["val_loss"]) < 1e-5 * abs(one["val_loss"]) and res[rank][3] == one["nested_val_metrics"]["t2v_metrics"]["R5"]
    assert np.array_equal(res[0][1], res[1][1])


# ---------------------------------------------------------------------------------------------------------------------
# (h) the selection kernel writes a batch straight into the captured graphs' input buffers
# ---------------------------------------------------------------------------------------------------------------------
def test_region_batcher_stages_into_graph_inputs_and_the_step_matches_the_copying_path():
    """``RegionBatcher.to_device(out=step.inputs)`` + ``step(step.inputs)``: the selection output is bit-identical to the
    free-standing call and the replayed steps return exactly what they return when the batch is copied in."""
    from demovlp_amd import functional as Fn
    from demovlp_amd.data import RegionBatcher
    F, R, B, NRAW = 8, 36, 2, 44
    rb = RegionBatcher(B, F, R, max_regions=64, device=DEV)

    def stage(seed):
        for b in range(B):
            for f in range(F):
                fr = syn.make_frame(seed + b, f, NRAW)
                rb.stage(b, f, fr["x"], fr["bbox"], fr["objects_conf"], (640.0, 360.0))

    ids, att = syn.caption_batch(B, first_sample=0)
    text = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(att).to(DEV)}
    losses = []
    for in_place in (False, True):
        Fn.SHADOWS.clear()
        model = build(F, R)
        opt = FusedAdamW(ParamArena(model), lr=1e-4)
        step = GraphedTrainStep(model, loss_head(), opt, warmup=2)
        got = []
        for it in range(6):                                     # 2 eager, the capture, 3 replays; a different clip set every step
            stage(20 + 3 * it)
            if in_place and step.inputs is not None:
                ins = step.inputs
                obj, mask, _ = rb.to_device(out=ins)
                assert obj.data_ptr() == ins["object"].data_ptr() and mask.data_ptr() == ins["object_mask"].data_ptr()
                for k, v in text.items():
                    ins["text"][k].copy_(v)
                data = ins
            else:
                obj, mask, _ = rb.to_device()
                data = {"text": text, "object": obj, "object_mask": mask}
            if it == 5:
                stage(20 + 3 * it)
                ref_obj, ref_mask, _ = rb.to_device()
                assert torch.equal(ref_obj, data["object"]) and torch.equal(ref_mask, data["object_mask"])
            got.append([float(t) for t in step(data)])
        losses.append(got)
    assert losses[0] == losses[1], (losses[0], losses[1])
    with pytest.raises(ValueError):
        ops.region_select(torch.zeros(1, 1, 4, 2048, device=DEV), torch.zeros(1, 1, 4, 4, device=DEV), torch.zeros(1, 1, 4, device=DEV),
                          torch.ones(1, 1, 2, device=DEV), 4, out=(torch.zeros(1, 1, 4, 2054, device=DEV), torch.zeros(1, 1, 5, device=DEV)))


def test_graph_step_recaptures_when_the_batch_shape_changes():
    """A loader's last batch is smaller (drop_last=False, base/base_data_loader.py:23-38): the replayed step must re-capture for the new
    shape -- and again when the full shape comes back -- and follow the eager sequence exactly."""
    from demovlp_amd import functional as Fn
    F, R = 8, 36
    full = to_dev(*golden_batch(F, R, 3))
    small = {"text": {k: v[:2].contiguous() for k, v in full["text"].items()}, "object": full["object"][:2].contiguous(),
             "object_mask": full["object_mask"][:2].contiguous()}
    seq = [full] * 4 + [small] * 3 + [full] * 2 + [small]        # both shapes captured (2 eager + capture each), then replayed alternately
    finals, losses = [], []
    for graphed in (False, True):
        Fn.SHADOWS.clear()
        model = build(F, R)
        arena = ParamArena(model)
        opt = FusedAdamW(arena, lr=1e-4)
        lf = loss_head()
        stepper = GraphedTrainStep(model, lf, opt, warmup=2) if graphed else None
        got = []
        for d in seq:
            out = stepper(d) if graphed else train_step(model, lf, opt, d)
            got.append(float(out[0]))
        torch.cuda.synchronize()
        if graphed:
            assert len(stepper._sets) == 2 and tuple(stepper.inputs["object"].shape) == tuple(small["object"].shape)
        finals.append(arena.flat_p.clone())
        losses.append(got)
    # bit for bit: no kernel of the step uses float atomics (the word-embedding gradient -- [CLS] / [SEP] sit in every caption -- is summed in
    # index order by the first row that holds the token), so eager and replayed sequences, and two runs of either, end on identical parameters
    assert losses[0] == losses[1], (losses[0], losses[1])
    assert torch.equal(finals[0], finals[1])


def test_training_steps_are_bit_reproducible_run_to_run():
    """Two runs of the same five bf16 steps (B = 4: four [CLS] rows, four [SEP] rows and repeated words collide in the word-embedding
    gradient) give identical losses and identical parameters."""
    from demovlp_amd import functional as Fn
    F, R, B = 8, 36, 4
    data = to_dev(*golden_batch(F, R, B))
    runs = []
    for _ in range(2):
        Fn.SHADOWS.clear()
        model = build(F, R, "bfloat16")
        model.set_text_dropout(0.1)                        # the Philox masks are a pure function of (seed, step, site, index)
        arena = ParamArena(model, bf16_shadow=True)
        opt = FusedAdamW(arena, lr=1e-4)
        lf = loss_head()
        ls = [float(train_step(model, lf, opt, data)[0]) for _ in range(5)]
        torch.cuda.synchronize()
        runs.append((ls, arena.flat_p.clone()))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_evaluation_between_replayed_steps_leaves_training_unchanged(dtype):
    """An epoch loop: replayed training steps, a validation pass (other batch sizes, no-grad kernels, the same workspaces and weight
    shadows), replayed steps again -- the parameters end where the same loop of eager steps ends, and the two validation results agree."""
    from demovlp_amd import functional as Fn
    from demovlp_amd.trainer import evaluate
    F, R, B = 8, 36, 2
    data = to_dev(*golden_batch(F, R, B))
    finals, vals = [], []
    for graphed in (False, True):
        Fn.SHADOWS.clear()
        model = build(F, R, dtype)
        arena = ParamArena(model, bf16_shadow=(dtype == "bfloat16"))
        opt = FusedAdamW(arena, lr=1e-4)
        lf = loss_head()
        stepper = GraphedTrainStep(model, lf, opt, warmup=2) if graphed else None
        run = (lambda: stepper(data)) if graphed else (lambda: train_step(model, lf, opt, data))
        for _ in range(4):
            run()
        res = evaluate(model, lf, _eval_batches(F, R, 8, 2))          # 16 pairs in batches of 8
        assert model.training
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        finals.append(arena.flat_p.clone())
        vals.append(res)
    assert torch.equal(finals[0], finals[1])
    # (the fused per-pair evaluation kernel of the bf16 path is not bit-reproducible from launch to launch: see the precision-knob test)
    assert abs(vals[0]["val_loss"] - vals[1]["val_loss"]) < 1e-5 * abs(vals[0]["val_loss"])
    assert np.allclose(vals[0]["o2t_sims"], vals[1]["o2t_sims"], rtol=1e-5, atol=1e-6)


def test_graph_capture_with_a_prefetching_loader_thread_running():
    """the loader thread stages, copies and waits on events while the step is being captured (and, with one captured
    set per shape, a capture can now come at any point of an epoch).  The capture must not be invalidated by that thread's HIP calls, and
    the steps must equal the same batches fed without a thread."""
    from demovlp_amd import functional as Fn
    from demovlp_amd.data import RegionBatcher, prefetching
    F, R, B, NRAW, STEPS = 8, 36, 2, 40, 7
    ids, att = syn.caption_batch(B, first_sample=0)
    text = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(att).to(DEV)}

    def batches(rb):
        for k in range(STEPS):
            for b in range(B):
                for f in range(F):
                    fr = syn.make_frame(50 + 5 * k + b, f, NRAW)
                    rb.stage(b, f, fr["x"], fr["bbox"], fr["objects_conf"], (640.0, 360.0))
            obj, mask, _ = rb.to_device()
            yield {"text": text, "object": obj, "object_mask": mask}

    losses = []
    for threaded in (False, True):
        Fn.SHADOWS.clear()
        model = build(F, R)
        opt = FusedAdamW(ParamArena(model), lr=1e-4)
        step = GraphedTrainStep(model, loss_head(), opt, warmup=2)
        rb = RegionBatcher(B, F, R, max_regions=64, device=DEV)
        it = prefetching(batches(rb), depth=3) if threaded else batches(rb)
        losses.append([float(step(d)[0]) for d in it])
        torch.cuda.synchronize()
    assert losses[0] == losses[1], (losses[0], losses[1])


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_replays_with_a_host_synchronisation_between_them(dtype):
    """A training loop that reads the loss every step (``loss.item()``) synchronises with the device between replays, so every replay
    starts on an idle GPU.  That is when a hipMemsetAsync captured inside the local-loss backward (a memset NODE) was not ordered against
    the kernels around it and the video-side gradients came back as garbage: the captured step holds kernel nodes only now, and the
    synchronised loop must end exactly where the eager loop ends."""
    from demovlp_amd import functional as Fn
    F, R, B = 8, 36, 2
    data = to_dev(*golden_batch(F, R, B))
    finals = []
    for graphed in (False, True, True):
        Fn.SHADOWS.clear()
        model = build(F, R, dtype)
        arena = ParamArena(model, bf16_shadow=(dtype == "bfloat16"))
        opt = FusedAdamW(arena, lr=1e-4)
        lf = loss_head()
        stepper = GraphedTrainStep(model, lf, opt, warmup=2) if graphed else None
        losses = []
        for _ in range(8):
            out = stepper(data) if graphed else train_step(model, lf, opt, data)
            losses.append(out[0].item())                      # host <- device: the next replay finds the GPU idle
            torch.cuda.synchronize()
        finals.append((losses, arena.flat_p.clone()))
    for ls, pf in finals[1:]:
        assert ls == finals[0][0], (ls, finals[0][0])
        assert torch.equal(pf, finals[0][1])


def _graph_dp_bf16_worker(rank, world, port, q, cut, grad_dtype="float32"):
    import traceback
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from demovlp_amd import functional as Fn
        F, R, B = 8, 36, 2
        runs = []
        for _ in range(2):            # the job twice, from scratch each time (fresh model, arena, optimizer state, graphs): run-to-run reproducibility
            Fn.SHADOWS.clear()
            model = build(F, R, "bfloat16")
            arena = ParamArena(model, bf16_shadow=True)
            opt = FusedAdamW(arena, lr=1e-4)
            stepper = GraphedTrainStep(model, loss_head(), opt, warmup=2, cut=cut, bucket_mb=64.0, grad_dtype=grad_dtype)
            losses = []
            for s in range(7):
                out = stepper(to_dev(*_np_batch(F, R, B, rank, s)))
                losses.append(float(out[0].item()))                   # host synchronisation every step: each replay starts on an idle device
                torch.cuda.synchronize()
            assert stepper.graph is not None
            runs.append((losses, arena.flat_p.double().cpu().numpy()[::211], float(arena.flat_p.double().sum().item())))
            del stepper, opt, arena, model
            torch.cuda.empty_cache()
        q.put((rank, runs))
    except BaseException:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("grad_dtype", ["float32", pytest.param("bfloat16", marks=pytest.mark.slow)])
def test_two_rank_bf16_graphed_step_with_host_sync_is_in_lock_step_and_reproducible(grad_dtype):
    """The data-parallel graph path in bf16 (three graphs, bucketed exchange, optimizer per bucket) with the loss read on the host every
    step: the two ranks stay bit-equal, every loss is finite, and a second run of the same job (fresh model, arena, optimizer state and
    graphs, in the same pair of processes) reproduces the first bit for bit.
    Gradient buckets cross as fp32 in the default run and as bf16 (cast kernel, bf16 sum, cast back) in the --runslow variant."""
    (_, r0), (_, r1) = _spawn(_graph_dp_bf16_worker, ((8, 4), grad_dtype))
    for (l0, p0, s0), (l1, p1, s1) in zip(r0, r1):                                   # the two ranks, run by run: in lock step
        assert np.array_equal(p0, p1) and s0 == s1
        assert np.isfinite(l0).all() and np.isfinite(l1).all() and np.isfinite(p0).all()
    for r in (r0, r1):                                                               # each rank, first run against second: bit for bit
        assert r[0][0] == r[1][0] and np.array_equal(r[0][1], r[1][1]) and r[0][2] == r[1][2]
