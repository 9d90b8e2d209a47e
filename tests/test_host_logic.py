"""Host-side logic that needs no GPU: state_dict schema, config factory, checkpoint helpers, loud failure on CPU,
and the data-parallel plumbing (AllGather_multi, GradReducer) on world_size-2 gloo."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from demovlp_amd import DemoVLPHipError, synthetic as syn
from demovlp_amd import config as cfgmod
import demovlp_amd.model as model_mod
import demovlp_amd.loss as loss_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {
    "n_gpu": 8,
    "pretrained_init": False,      # synthetic runs: the factory fills constructor parameters from top-level keys (parse_config_dist_multi.py:88-92)
    "arch": {"type": "ObjectRelation", "args": {
        "object_params": {"model": "", "input_objects": False, "object_num": 30, "num_frames": 1, "time_module": None},
        "text_params": {"model": "pretrained/distilbert-base-uncased", "pretrained": True, "input": "text", "two_outputs": True},
        "projection": "minimal", "load_checkpoint": ""}},
    "loss": {"type": "GlobalLocalLoss", "args": {"use_local": True, "use_global": True, "coef": 1.0, "focal_type": "equal"}},
}


def test_config_factory_builds_dropin_modules():
    m = cfgmod.initialize(CFG, "arch", model_mod)
    l = cfgmod.initialize(CFG, "loss", loss_mod)
    assert isinstance(m, model_mod.ObjectRelation) and isinstance(l, loss_mod.GlobalLocalLoss)
    assert m.segments == 1 and m.projection_dim == 256
    assert l.local_loss.focal_type == "equal" and l.global_loss.temperature == 0.05


@pytest.mark.parametrize("F,R", [(1, 30), (8, 36)])
def test_state_dict_schema_matches_reference(F, R):
    cfg = json.loads(json.dumps(CFG))
    cfg["arch"]["args"]["object_params"].update(object_num=R, num_frames=F)
    m = cfgmod.initialize(cfg, "arch", model_mod)
    sd, schema = m.state_dict(), syn.state_dict_schema(F, R)
    assert len(sd) == 280 and set(sd) == set(schema)
    assert all(tuple(sd[k].shape) == tuple(schema[k]) for k in schema)
    assert sum(v.numel() for v in sd.values()) == sum(int(np.prod(s)) for s in schema.values())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(F, R).items()}, strict=True)


def test_reference_argument_errors():
    with pytest.raises(NotImplementedError):
        model_mod.ObjectRelation({"object_num": 30, "num_frames": 1, "time_module": None}, {"model": "", "pretrained": False})
    with pytest.raises(NotImplementedError):
        model_mod.ObjectRelation({"object_num": 30, "num_frames": 1, "time_module": None}, {"model": "", "pretrained": True},
                                 projection="full", pretrained_init=False)


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = cfgmod.initialize(CFG, "arch", model_mod)
    data = {"text": {"input_ids": torch.zeros(2, 100, dtype=torch.long), "attention_mask": torch.ones(2, 100, dtype=torch.long)},
            "object": torch.zeros(2, 1, 30, 2054), "object_mask": torch.ones(2, 1, 30)}
    with pytest.raises(DemoVLPHipError):
        m(data)
    with pytest.raises(DemoVLPHipError):
        model_mod.sim_matrix(torch.randn(4, 256), torch.randn(4, 256))
    with pytest.raises(DemoVLPHipError):
        loss_mod.RWALoss().get_sim(torch.randn(2, 30, 256), torch.randn(2, 99, 256), torch.zeros(2, 30), None, torch.zeros(2, 99))


def test_checkpoint_helpers(tmp_path):
    m = cfgmod.initialize(CFG, "arch", model_mod)
    sd = {"module." + k: v.clone() for k, v in m.state_dict().items()}
    fixed = model_mod.state_dict_data_parallel_fix(sd, m.state_dict())
    assert set(fixed) == set(m.state_dict())
    # temporal-embed inflation: load a 1-frame checkpoint into a 4-frame model
    cfg4 = json.loads(json.dumps(CFG))
    cfg4["arch"]["args"]["object_params"]["num_frames"] = 4
    ck = tmp_path / "ck.pth"
    src = m.state_dict()
    src["object_model.temporal_embed"] = torch.full((1, 1, 768), 0.5)
    torch.save({"state_dict": {"module." + k: v for k, v in src.items()}}, ck)
    cfg4["arch"]["args"]["load_checkpoint"] = str(ck)
    m4 = cfgmod.initialize(cfg4, "arch", model_mod)
    te = m4.state_dict()["object_model.temporal_embed"]
    assert te.shape == (1, 4, 768) and float(te[0, 0, 0]) == 0.5 and float(te[0, 1:].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from demovlp_amd.trainer import AllGather_multi, GradReducer, ParamArena

    class Args:
        pass
    args = Args()
    args.world_size, args.rank = world, rank
    # AllGather_multi: forward = cat over ranks, backward = this rank's slice, no reduction (trainer_dist.py:13-31)
    x = (torch.arange(6, dtype=torch.float32).reshape(3, 2) + 100 * rank).requires_grad_(True)
    g = AllGather_multi.apply(x, world, args)
    w = torch.arange(g.numel(), dtype=torch.float32).reshape(g.shape)
    (g * w).sum().backward()
    ok_gather = torch.equal(g.detach()[3 * rank:3 * rank + 3], x.detach()) and g.shape == (3 * world, 2) \
        and torch.equal(x.grad, w[3 * rank:3 * rank + 3])
    # gather_embeddings (the opt-in cross-GPU negatives): every rank sees all ranks' embeddings in rank order, masks / lengths
    # are gathered without gradient, and the backward of each embedding is this rank's slice (no reduce-scatter)
    from demovlp_amd.trainer import gather_embeddings
    emb = {k: (torch.randn(3, *shape) + 10 * rank).requires_grad_(True) for k, shape in
           (("global_text_embeddings", (4,)), ("local_text_embeddings", (5, 4)), ("global_object_embeddings", (4,)), ("local_object_embeddings", (6, 4)))}
    emb["object_mask"] = torch.full((3, 6), float(rank))
    gout, glen, gmask = gather_embeddings(emb, torch.full((3,), rank + 7), torch.full((3, 5), -100.0 * rank), args)
    ok_gather &= all(gout[k].shape[0] == 3 * world and torch.equal(gout[k][3 * rank:3 * rank + 3].detach(), emb[k].detach()) for k in emb)
    ok_gather &= glen.tolist() == [7] * 3 + [8] * 3 and gmask.shape == (6, 5) and float(gmask[3:].max()) == -100.0
    ok_gather &= float(gout["object_mask"][3:].min()) == 1.0 and not gout["object_mask"].requires_grad
    tot = sum((gout[k] * (1.0 + torch.arange(gout[k].shape[0], dtype=torch.float32).reshape(-1, *[1] * (gout[k].dim() - 1)))).sum()
              for k in emb if k != "object_mask")
    tot.backward()
    ok_gather &= all(torch.equal(emb[k].grad, (1.0 + torch.arange(3 * rank, 3 * rank + 3, dtype=torch.float32)).reshape(-1, *[1] * (emb[k].dim() - 1)).expand_as(emb[k]))
                     for k in emb if k != "object_mask")
    # GradReducer over a flat arena: bucketed in-place all-reduce, never-used tensors excluded
    torch.manual_seed(0)
    from demovlp_amd.functional import _grad_buf

    class ArenaLinear(torch.autograd.Function):
        """CPU stand-in for the HIP layer nodes: backward writes the weight/bias gradients into the arena views and
        hands those views to autograd (which adopts them), exactly as functional.py does on the GPU."""

        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            ctx.b = b
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            gw, gb = _grad_buf(w), _grad_buf(ctx.b)
            gw.copy_(dy.t() @ x)
            gb.copy_(dy.sum(0))
            return dy @ w, gw, gb

    net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 8))
    unused = torch.nn.Parameter(torch.ones(5))
    net.register_parameter("unused", unused)
    arena = ParamArena(net, device="cpu")
    red = GradReducer(arena, bucket_mb=0.01)
    outs = []
    adopted = True
    for step in range(3):
        for p in net.parameters():
            p.grad = None
        red.begin()
        h = torch.full((4, 64), float(rank + 1 + step))
        for lin in net:
            h = ArenaLinear.apply(h, lin.weight, lin.bias)
        h.sum().backward()
        adopted &= all(p.grad.data_ptr() == p._dvlp_grad_view.data_ptr() for p in net.parameters() if p is not unused)
        scale = red.finish()
        outs.append((arena.flat_g.clone(), scale))
    # reference: sum over ranks of the per-rank gradients
    ref = []
    for step in range(3):
        tot = None
        for r in range(world):
            n2 = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 8))
            n2.load_state_dict({k: v for k, v in net.state_dict().items() if k != "unused"})
            n2(torch.full((4, 64), float(r + 1 + step))).sum().backward()
            g2 = dict(n2.named_parameters())          # arena order (matrices first, vectors last), not module order
            flat = torch.cat([g2[n].grad.reshape(-1) for n in arena.names if n != "unused"])
            tot = flat if tot is None else tot + flat
        ref.append(tot)
    ok_red = True
    for (flat, scale), r in zip(outs, ref):
        got = torch.cat([flat[o:o + p.numel()] for p, o in zip(arena.params, arena.offsets) if p is not unused])
        ok_red &= bool(torch.allclose(got, r, rtol=1e-5, atol=1e-5)) and scale == 1.0 / world
        ui = [i for i, p in enumerate(arena.params) if p is unused][0]
        lo, hi = arena.slice_of(ui)
        ok_red &= float(flat[lo:hi].abs().max()) == 0.0
    ok_red &= len(red.buckets) > 1 and red.expected is not None and adopted
    q.put((rank, ok_gather, ok_red))
    dist.destroy_process_group()


def test_data_parallel_plumbing_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(g and r for _, g, r in res), res


def test_retrieval_metrics_match_reference_golden():
    """demovlp_amd.metric against vectors produced by the reference's model/metric.py (ties included)."""
    from demovlp_amd import metric
    g = np.load(os.path.join(ROOT, "tests", "golden", "g7_metrics.npz"))
    keys = ("R1", "R5", "R10", "R50", "MedR", "MeanR", "geometric_mean_R1-R5-R10")
    for tag in ("sq64", "rect"):
        sims = g[tag + "_sims"]
        for name, fn in (("t2v", metric.t2v_metrics), ("v2t", metric.v2t_metrics)):
            got = fn(sims.copy())
            assert np.allclose([got[k] for k in keys], g[f"{tag}_{name}"], rtol=1e-12, atol=1e-12), (tag, name)


# ------------------------------------------------------------------------------------------------------------------
# input side (demovlp_amd/data.py)
# ------------------------------------------------------------------------------------------------------------------
def test_shard_indices_equal_torch_distributed_sampler():
    """Per-rank index lists: identical to DistributedSampler(shuffle, drop_last=True) after set_epoch, ragged sizes included."""
    from torch.utils.data.distributed import DistributedSampler
    from demovlp_amd.data import shard_indices
    for n in (1000, 1003, 17, 8, 5):
        for world in (1, 2, 8):
            for shuffle in (True, False):
                for epoch in (0, 3):
                    for rank in range(world):
                        s = DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=shuffle, drop_last=True)
                        s.set_epoch(epoch)
                        assert list(s) == shard_indices(n, world, rank, epoch, shuffle).tolist(), (n, world, rank, shuffle, epoch)


def test_frame_sampling_and_npz_schema(tmp_path):
    """Frame choice (base/base_dataset.py:82-101) and the per-frame .npz round trip in the reference's schema."""
    import random
    from demovlp_amd import synthetic as syn
    from demovlp_amd.data import read_frame_npz, sample_frame_indices, alternate
    assert sample_frame_indices(8, 8, "rand") == list(range(8))
    assert sample_frame_indices(4, 16, "uniform") == [1, 5, 9, 13]            # midpoints of (0,3) (4,7) (8,11) (12,15)
    assert sample_frame_indices(8, 3, "uniform") == [0, 1, 2]                 # fewer files than segments: min(...) intervals
    for seed in range(20):
        idx = sample_frame_indices(8, 30, "rand", random.Random(seed))
        iv = np.linspace(0, 30, 9).astype(int)
        assert idx == sorted(idx) and all(iv[i] <= idx[i] < iv[i + 1] - 1 for i in range(8))    # the reference never picks the last frame of an interval
    fr = syn.make_frame(5, 2, 28)
    syn.save_frame_npz(str(tmp_path / "2.npz"), fr)
    x, bbox, conf, wh = read_frame_npz(str(tmp_path / "2.npz"))
    assert np.array_equal(x, fr["x"]) and np.array_equal(bbox, fr["bbox"]) and np.array_equal(conf, fr["objects_conf"]) and wh == (640.0, 360.0)
    assert list(alternate([[1, 2, 3], ["a", "b"]])) == [(0, 1), (1, "a"), (0, 2), (1, "b")]


def test_adjust_learning_rate_reproduces_the_reference_quirk():
    """trainer/trainer_dist.py:97-102: lr := learning_rate1 x 0.1 per passed milestone, whatever the config's lr was."""
    from demovlp_amd.trainer import adjust_learning_rate

    class Opt:
        param_groups = [{"lr": 1e-5}, {"lr": 3e-5}]

    class Args:
        learning_rate1, schedule = 2e-4, [3, 6]
    for epoch, want in ((1, 2e-4), (2, 2e-4), (3, 2e-5), (5, 2e-5), (6, 2e-6), (9, 2e-6)):
        assert abs(adjust_learning_rate(Opt, epoch, Args) - want) < 1e-18
        assert all(abs(g["lr"] - want) < 1e-18 for g in Opt.param_groups)


def test_fused_adamw_state_dict_uses_the_reference_checkpoint_layout():
    """base/base_trainer.py:185-192 stores transformers.AdamW.state_dict(): {'state': {i: {step, exp_avg, exp_avg_sq}},
    'param_groups': [{..., 'params': [0..n-1]}]}, i = position among the trainable parameters in model.parameters() order --
    not the arena's (matrices first) order -- and no entry for tensors that never had a gradient."""
    from demovlp_amd.trainer import FusedAdamW, ParamArena
    net = torch.nn.Sequential(torch.nn.Linear(8, 4), torch.nn.LayerNorm(4), torch.nn.Linear(4, 2))
    order = [n for n, _ in net.named_parameters()]                  # 0.weight 0.bias 1.weight 1.bias 2.weight 2.bias
    arena = ParamArena(net, device="cpu")
    assert [arena.names[i] for i in arena.model_order] == order and arena.names[:2] == ["0.weight", "2.weight"]
    opt = FusedAdamW(arena, lr=3e-4)
    opt.step_count = 5
    touched = [i for i, n in enumerate(arena.names) if n != "1.bias"]
    opt._ever = set(touched)
    for i in touched:
        lo, hi = arena.slice_of(i)
        opt.m[lo:hi] = float(i + 1)
        opt.v[lo:hi] = float(10 * (i + 1))
    sd = opt.state_dict()
    assert sorted(sd["state"]) == [0, 1, 2, 4, 5] and sd["param_groups"][0]["params"] == list(range(6))
    assert sd["param_groups"][0]["lr"] == 3e-4 and sd["param_groups"][0]["betas"] == (0.9, 0.999) and sd["param_groups"][0]["correct_bias"] is True
    for k, st in sd["state"].items():
        i = arena.names.index(order[k])
        assert st["step"] == 5 and st["exp_avg"].shape == arena.params[i].shape
        assert float(st["exp_avg"].mean()) == i + 1 and float(st["exp_avg_sq"].mean()) == 10 * (i + 1)
    net2 = torch.nn.Sequential(torch.nn.Linear(8, 4), torch.nn.LayerNorm(4), torch.nn.Linear(4, 2))
    opt2 = FusedAdamW(ParamArena(net2, device="cpu"), lr=1.0)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 5 and opt2.param_groups[0]["lr"] == 3e-4 and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    assert opt2._ever == opt._ever


def test_two_graph_exchange_plan_covers_the_arena_once():
    """GraphedTrainStep.plan_exchange: the ranges exchanged behind the first piece of a cut backward (text tower + object blocks >= cut)
    and the ranges exchanged at the end are disjoint, ordered, and together cover the arena exactly; the early ones hold only tensors
    whose gradients the first piece finishes."""
    from types import SimpleNamespace
    from demovlp_amd.trainer import GraphedTrainStep
    names = (["text_model.embeddings.word_embeddings.weight"] + [f"text_model.transformer.layer.{l}.{w}.weight" for l in range(2) for w in ("q_lin", "ffn")]
             + ["object_model.cls_token", "object_model.temporal_embed"] + [f"object_model.blocks.{b}.{w}.weight" for b in range(4) for w in ("attn.qkv", "mlp.fc1")]
             + ["object_model.object_embedding.weight", "object_model.proj.weight", "txt_proj.1.weight"])
    n_matrix = len(names)
    names += ["text_model.some.bias", "object_model.blocks.0.norm1.weight"]
    sizes = [300000 + 64 * i for i in range(len(names))]
    offs, o = [], 0
    for z in sizes:
        offs.append(o)
        o += (z + 63) // 64 * 64
    arena = SimpleNamespace(names=names, offsets=offs, total=o, n_matrix=n_matrix, ALIGN=64, params=[SimpleNamespace(numel=lambda z=z: z) for z in sizes])
    early, late = GraphedTrainStep.plan_exchange(arena, 2)
    assert len(early) == 2                                       # the text tower, and object blocks 2..3
    spans = sorted(list(early) + list(late))
    assert spans[0][0] == 0 and spans[-1][1] == arena.total
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for lo, hi in early:
        inside = [n for n, off in zip(names, offs) if lo <= off < hi]
        assert inside and all(n.startswith("text_model.") or int(n.split(".")[2]) >= 2 for n in inside)
        assert all(names.index(n) < n_matrix for n in inside)
    # several cuts: piece 0 = text tower + blocks >= 3, piece 1 = block 2, piece 2 = block 1, last = the rest (block 0, prologue, heads, vectors)
    pieces = GraphedTrainStep.plan_exchange(arena, (3, 2, 1))
    assert len(pieces) == 4 and all(pieces)
    spans = sorted(r for runs in pieces for r in runs)
    assert spans[0][0] == 0 and spans[-1][1] == arena.total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    want = [lambda n: n.startswith("text_model.") or int(n.split(".")[2]) >= 3, lambda n: n.split(".")[2] == "2", lambda n: n.split(".")[2] == "1"]
    for k in range(3):
        for lo, hi in pieces[k]:
            inside = [n for n, off in zip(names, offs) if lo <= off < hi]
            assert inside and all(want[k](n) for n in inside) and all(names.index(n) < n_matrix for n in inside)
    last = [n for lo, hi in pieces[3] for n, off in zip(names, offs) if lo <= off < hi]
    assert "object_model.blocks.0.attn.qkv.weight" in last and "txt_proj.1.weight" in last and "text_model.some.bias" in last


def test_graphed_step_host_bookkeeping_bucket_alignment_and_lru():
    """(a) an odd ``bucket_mb`` must not give range starts adamw_range_dev rejects -- the bucket is rounded to the arena's
    alignment; (b) captured sets are evicted least-recently-USED, not first-captured; (c) ``inputs_for`` names a shape's own buffers and is
    None while that shape is still in its eager warm-up.  Pure host logic: no kernel is called."""
    from types import SimpleNamespace
    from demovlp_amd.trainer import GraphedTrainStep
    arena = SimpleNamespace(ALIGN=64, total=10 * 64 * 1000)
    opt = SimpleNamespace(arena=arena)
    for mb in (0.3, 1.5e-3, 64.0, 1e-9):
        st = GraphedTrainStep(SimpleNamespace(), None, opt, bucket_mb=mb)
        assert st.bucket >= 64 and st.bucket % 64 == 0, (mb, st.bucket)
        pieces = list(st._pieces([(0, arena.total)]))
        assert pieces[0][0] == 0 and pieces[-1][1] == arena.total and all(lo % 64 == 0 for lo, _ in pieces)
        assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
    st = GraphedTrainStep(SimpleNamespace(), None, opt)
    st.max_shapes, st.warmup = 2, 1
    captured, replays = [], []

    def batch(n):
        return {"text": {"input_ids": torch.zeros(n, 4, dtype=torch.long), "attention_mask": torch.ones(n, 4)}, "object": torch.zeros(n, 2, 3, 8),
                "object_mask": torch.ones(n, 2)}
    st._eager = lambda d: ("eager", d["object"].shape[0])

    def fake_capture(d):
        n = d["object"].shape[0]
        captured.append(n)
        st.graphs = [SimpleNamespace(replay=lambda n=n: replays.append(n))]
        st.static = {"text": {k: v.clone() for k, v in d["text"].items()}, "object": d["object"].clone(), "object_mask": d["object_mask"].clone()}
        st.out = (torch.zeros(()),)
        st.shape_key = st._key(d)
    st._capture = fake_capture
    st.opt = SimpleNamespace(_sync_hyper=lambda s: None, replayed=lambda: None, arena=arena)
    assert st.inputs_for(batch(4)) is None
    assert st(batch(4))[0] == "eager" and st.inputs_for(batch(4)) is None          # warm-up call: no capture yet
    st(batch(4))                                                                    # captured + replayed
    assert captured == [4] and st.inputs_for(batch(4))["object"].shape[0] == 4
    st(batch(3)); st(batch(3))                                                      # second shape
    assert captured == [4, 3] and st.inputs_for(batch(3))["object"].shape[0] == 3 and st.inputs_for(batch(4))["object"].shape[0] == 4
    st(batch(4))                                                                    # touch shape 4: shape 3 is now the least recently used
    st(batch(2)); st(batch(2))                                                      # third shape evicts ... shape 3, not the main shape 4
    assert captured == [4, 3, 2]
    assert st.inputs_for(batch(3)) is None and st.inputs_for(batch(4)) is not None and st.inputs_for(batch(2)) is not None
    n_before = len(captured)
    st(batch(4))
    assert len(captured) == n_before                                               # still captured: replayed, not re-captured


def _bf16_bucket_worker(rank, world, port, q):
    import traceback
    from types import SimpleNamespace
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from demovlp_amd.trainer import GraphedTrainStep
        n = 64 * 300
        gen = torch.Generator().manual_seed(100 + rank)
        out = {}
        for dtype in ("float32", "bfloat16"):
            g = torch.randn(n, generator=torch.Generator().manual_seed(100 + rank)) * 1e-3
            arena = SimpleNamespace(ALIGN=64, total=n, flat_g=g)
            st = GraphedTrainStep(SimpleNamespace(), None, SimpleNamespace(arena=arena), bucket_mb=64 * 40 * 4 / 2 ** 20, grad_dtype=dtype)
            st.world, st.collective = world, True
            st._exchange_and_update([(0, 64 * 100), (64 * 100, n)])
            out[dtype] = g.clone()
        q.put((rank, out["float32"].numpy(), out["bfloat16"].numpy()))
    except BaseException:  # noqa: BLE001
        q.put((rank, traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_bf16_gradient_buckets_gloo_world2():
    """GraphedTrainStep(grad_dtype='bfloat16'), host path: every bucket is summed across the ranks in bf16 and lands back in the fp32
    gradient buffer -- identical on both ranks, within bf16 rounding of the fp32 exchange (the optimizer's inputs stay fp32 tensors)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    (_, f0, h0), (_, f1, h1) = res
    assert np.array_equal(f0, f1) and np.array_equal(h0, h1)                       # ranks in lock step either way
    want = sum((torch.randn(64 * 300, generator=torch.Generator().manual_seed(100 + r)) * 1e-3) for r in range(2)).numpy()
    assert np.abs(f0 - want).max() <= 1e-9
    rel = np.abs(h0 - want).max() / np.abs(want).max()
    assert 0 < rel < 2 ** -7, rel                                                  # bf16 inputs + a bf16 sum: three roundings of 2^-9


def _rs_ag_worker(rank, world, port, q):
    import traceback
    import warnings
    from types import SimpleNamespace
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from demovlp_amd.trainer import GraphedTrainStep
        n = 64 * 300 + 64                                  # buckets of 64 * 40 elements: with world = 3 none of them divides by the world size
        out = {}
        for xch in ("all_reduce", "rs_ag"):
            for dtype in ("float32", "bfloat16"):
                g = torch.randn(n, generator=torch.Generator().manual_seed(100 + rank)) * 1e-3
                arena = SimpleNamespace(ALIGN=64, total=n, flat_g=g)
                st = GraphedTrainStep(SimpleNamespace(), None, SimpleNamespace(arena=arena), bucket_mb=64 * 40 * 4 / 2 ** 20, grad_dtype=dtype, exchange=xch)
                st.world, st.collective = world, True
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    st._exchange_and_update([(0, 64 * 100), (64 * 100, n - 64)])
                    st._exchange_and_update([(n - 64, n)])          # a second piece: one 64-element bucket
                out[(xch, dtype)] = (g.clone().numpy(), sorted(st.exchange_used), len(w))
        q.put((rank, out))
    except BaseException:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rs_ag_shard_walk_gloo(world):
    """GraphedTrainStep(exchange='rs_ag') with more than one rank (the in-place shard views g[lo + r sh : lo + (r + 1) sh] with
    r > 0 and the scatter -> cast -> gather order had only ever run with one rank, where both collectives are identities).  gloo has no
    reduce_scatter_tensor, so the scatter is emulated (all_reduce of a copy, own shard kept, the rest of the bucket poisoned with NaN as the
    native in-place form leaves it undefined) -- the shard offsets, the divisible-prefix / remainder split (world = 3 does not divide the
    64-element bucket alignment) and the gather are the code a node runs.  fp32: bit-equal to the all_reduce form; bf16 buckets: the
    scatter stays fp32 and only the gather rounds, so the result is bf16(fp32 sum) exactly.  The form taken is recorded, the fallback warns once."""
    from demovlp_amd.trainer import GraphedTrainStep
    assert GraphedTrainStep.shard_split(128, 128 + 2560, 3) == (128 + 2559, 853) and GraphedTrainStep.shard_split(0, 64, 8) == (64, 8)
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rs_ag_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    n = 64 * 300 + 64
    want = sum((torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) * 1e-3) for r in range(world))
    for key in (("all_reduce", "float32"), ("rs_ag", "float32"), ("all_reduce", "bfloat16"), ("rs_ag", "bfloat16")):
        for r in range(1, world):
            assert np.array_equal(res[0][1][key][0], res[r][1][key][0]), key             # ranks in lock step, no NaN left behind
    ar, rs = res[0][1][("all_reduce", "float32")], res[0][1][("rs_ag", "float32")]
    assert np.isfinite(rs[0]).all() and np.abs(rs[0] - want.numpy()).max() <= 2e-9
    if world == 2:
        assert np.array_equal(ar[0], rs[0])                                               # two summands: no order to differ in
    rs16 = res[0][1][("rs_ag", "bfloat16")]
    mid = [GraphedTrainStep.shard_split(lo, hi, world)[0] for lo, hi in ((0, 2560), (2560, 5120))]
    assert np.array_equal(rs16[0][:mid[0]], torch.from_numpy(rs[0][:mid[0]]).to(torch.bfloat16).float().numpy())   # ONE rounding, of the fp32 sum
    assert rs[1] == ["rs_ag (emulated: all_reduce + all_gather)"] and ar[1] == ["all_reduce"]
    assert rs[2] == 1 and ar[2] == 0                                                      # one warning per step object


def test_no_undefined_names_anywhere_in_the_tree():
    """Round 4's GPU suite went red on `grad_dtype=grad_dtype` inside a worker that has no such name -- a NameError only the GPU box could
    raise.  Python's own symbol tables find that class of slip in a second: every name a function loads must be local, enclosing,
    module-level, imported or a builtin, in tests/, the package, tools/, oracle/, bench.py and __graft_entry__.py."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import undefined_names as un
    # the scanner must see the very slip it was written for
    slip = ("def outer():\n    grad_dtype = 'float32'\n    return grad_dtype\n\n"
            "def worker(rank):\n    return dict(rank=rank, grad_dtype=grad_dtype)\n")
    assert un.scan_source(slip, "slip.py") == [("slip.py", "worker", "grad_dtype")]
    assert un.scan_source("import os\nX = 1\ndef f(a):\n    global Y\n    Y = a\n    return [os.sep, X, Y, len(a)] + [b for b in a]\n") == []
    files = list(un.tree_files())
    assert len(files) > 40 and any(f.endswith("test_gpu_round3.py") for f in files)
    bad = [b for f in files for b in un.scan(f)]
    assert not bad, "\n".join("%s: %s: undefined name %r" % b for b in bad)


def test_every_test_is_mapped_to_a_survey_row():
    """tests/rows.py is the row-by-row index of the suite: every test function must name the row(s) it covers, every row
    must be covered by at least one GPU test where it is a device path, and the map must not name functions that no longer exist."""
    import ast
    import glob
    import rows
    have = {}
    for path in glob.glob(os.path.join(ROOT, "tests", "test_*.py")):
        src = open(path).read()
        gpu_file = any(line.startswith("pytestmark = pytest.mark.gpu") for line in src.splitlines())
        for node in ast.parse(src).body:
            if isinstance(node, ast.FunctionDef) and node.name.startswith("test_"):
                have.setdefault(node.name, []).append(gpu_file)
    assert not sorted(set(have) - set(rows.ROWS)), "tests missing from tests/rows.py"
    assert not sorted(set(rows.ROWS) - set(have)), "tests/rows.py names functions that do not exist"
    assert all(set(v.split()) <= set(rows.ALL_ROWS) and v.split() for v in rows.ROWS.values())
    for r in rows.ALL_ROWS:
        fns = [f for f, v in rows.ROWS.items() if r in v.split()]
        assert fns, r
        if r not in ("b", "c"):                                      # every path row has at least one test that runs on the device
            assert any(any(have[f]) for f in fns), r
