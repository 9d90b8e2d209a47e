#!/usr/bin/env python
"""Throughput of the DemoVLP cross-modal hot path on MI355X: video-text pairs/s, fwd + bwd + optimizer step.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a torchrun environment: starts its own N ranks)

A step = trainer/trainer_dist.py:144-171 on one synthetic batch already resident in HBM: ObjectRelation forward (region
transformer + DistilBERT), sim_matrix, GlobalLocalLoss (NT-Xent-style global + region<->word local loss, focal gate
'equal'), backward, fused HF-AdamW.  Workload = BASELINE.json configs[1]: 8 frames x 36 regions x 2048-d synthetic region
features + random captions, per-GPU batch 64, bf16 MFMA / fp32 accumulate, random-init (closed-form) weights.
N > 1: pure data parallel (weak scaling), gradient all-reduce over RCCL overlapped with backward; per-rank local
negatives exactly as the reference trains.

Prints ONE JSON line (rank 0).  ``roofline`` is for the dominant kernel (the MFMA GEMM family): algorithmic flops of
all its launches / their summed duration, measured with HIP events on the launch stream inside the timed region.
``cpu_baseline`` times the CPU oracle (oracle/restatement.py, a port proven equal to the reference by tests/golden) on
the host cores, on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
TRAFFIC_FILES = ("r6_final_gemm_hbm_traffic_pmc.json", "r5_final_gemm_hbm_traffic_pmc.json", "r4_final_gemm_hbm_traffic_pmc.json", "r3_final_gemm_hbm_traffic_pmc.json", "r2_final_gemm_hbm_traffic_pmc.json", "r1_final_gemm_hbm_traffic_pmc.json")


def csrc_sha():
    """sha256 over the kernel sources of the running tree (demovlp_amd/csrc/*): the committed PMC traffic figure is only reported
    when it was measured on exactly these kernels (tools/profile_round.sh records the same hash beside the counters)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "demovlp_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def flops_per_pair(B, F, R, W=99, Lt=100):
    """Algorithmic fwd+bwd FLOPs per video-text pair; bwd = 2 x fwd."""
    N = 1 + F * R
    enc = 2 * F * R * 2054 * 768 + 12 * (2 * N * (768 * 2304 + 768 * 768 + 2 * 768 * 3072) + 4 * F * R * (R + 1) * 768 + 4 * N * 768) \
        + 2 * N * 768 * 256
    txt = 6 * (2 * Lt * (4 * 768 * 768 + 2 * 768 * 3072) + 4 * Lt * Lt * 768) + 2 * Lt * 768 * 256
    loc = B * 3 * 2 * F * R * W * 256
    return 3.0 * (enc + txt + loc), 3.0 * enc


def text_flops_per_pair(Lt=100):
    """DistilBERT + txt_proj forward + backward per caption of Lt tokens (the `txt` term above x 3): 26.15 GFLOP at Lt = 100."""
    return 3.0 * (6 * (2 * Lt * (4 * 768 * 768 + 2 * 768 * 3072) + 4 * Lt * Lt * 768) + 2 * Lt * 768 * 256)


def usable_cores(cap=64):
    """Cores this process may really use: affinity mask, clipped by the cgroup CPU quota (a 256-thread pool inside an
    8-core quota runs two orders of magnitude slower than 8 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _cpu_leg(B, F, R, budget_s, max_steps):
    """oracle.train_step (fwd + loss + bwd, fp32, torch CPU) at per-step batch B: (pairs/s, description)."""
    import torch
    from demovlp_amd import synthetic as syn
    from oracle import restatement as orc
    p = orc.params_from_numpy(syn.fill_state_dict(F, R), requires_grad=True)
    obj, mask = syn.fast_region_batch(B, F, R)
    ids, att = syn.caption_batch(B)
    args = (torch.from_numpy(ids), torch.from_numpy(att), torch.from_numpy(obj), torch.from_numpy(mask))
    tw = time.perf_counter()
    orc.train_step(p, *args)                                   # warm-up
    tw = time.perf_counter() - tw
    if tw > budget_s:                                          # a single step already exhausts the budget: report it
        return B / tw, f"B={B}: 1 step (the warm-up itself, {tw:.1f} s)"
    n, t0 = 0, time.perf_counter()
    while True:
        for v in p.values():
            v.grad = None
        orc.train_step(p, *args)
        n += 1
        dt = time.perf_counter() - t0
        if dt + dt / n > budget_s or n >= max_steps:
            break
    return B * n / dt, f"B={B}: {n} steps after 1 warm-up ({dt:.1f} s)"


def cpu_baseline(F, R):
    """The oracle's full train step on the host cores at B=2 (BASELINE.json configs[0]) and B=16,
    about 10 s of CPU work each.  ``value`` is the better of the two rates."""
    import torch
    cores = usable_cores()
    torch.set_num_threads(cores)
    v2, d2 = _cpu_leg(2, F, R, 8.0, 6)
    v16, d16 = _cpu_leg(16, F, R, 12.0, 2)
    return dict(value=round(max(v2, v16), 3), unit="pairs/s", cores=cores, kind="port", cpu=cpu_model_name(),
                pairs_per_s_b2=round(v2, 3), pairs_per_s_b16=round(v16, 3),
                sample=f"oracle.train_step (fwd+loss+bwd, fp32, torch CPU, {cores} threads) at F={F}, R={R}; {d2}; {d16}")


def spawn_ranks(n):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh child ranks (one per GPU, RCCL rendezvous on
    127.0.0.1) exactly as the driver's multi-GPU command would, relay their output, exit with their code.  The parent never
    touches the GPU (train_dist_multi.py:33-38, 159-164 is the launch contract this mirrors)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def time_object_tower(model, data, steps, dist_sync, graph=True, part="object"):
    """ObjectTransformer alone, forward + backward (weight gradients included, no optimizer): seconds per pass.  This is the
    quantity BASELINE.json's north_star prices at >= 40 % of the bf16 MFMA peak (151.55 GFLOP per pair at F=8, R=36).
    ``graph``: the pass is captured once and replayed, like the step itself (the eager loop's ~350 launches cost the host about as
    long as the device needs for them, so an eager figure is partly a host figure); falls back to eager launches if capture fails.
    ``part``: "object" (default), "text" (DistilBERT + txt_proj alone, forward + backward) or "loss" (sim_matrix + GlobalLocalLoss on fixed
    embeddings, forward + backward) -- the other two terms of the step, timed the same way for the breakdown in the bench line.
    Returns (seconds per pass, "hipGraph replay" | "eager")."""
    import torch
    from demovlp_amd import functional as Fn, ops
    obj, mask = data["object"], data["object_mask"]
    state = {"dy": None}
    if part == "loss":
        from demovlp_amd.loss import GlobalLocalLoss
        from demovlp_amd.model import sim_matrix
        lf = GlobalLocalLoss(use_local=True, use_global=True, coef=1.0, focal_type="equal")
        with torch.no_grad():
            emb = model(data)
        leaves = {k: v.detach().clone().requires_grad_(v.is_floating_point() and k != "object_mask") for k, v in emb.items()}
        att = data["text"]["attention_mask"]
        tmask, tlen = (att[:, 1:].contiguous() - 1.0) * 100.0, att.sum(1)

    def one():
        if part == "loss":
            for v in leaves.values():
                v.grad = None
            gs = sim_matrix(leaves["global_text_embeddings"], leaves["global_object_embeddings"])
            loss = lf(gs, leaves["local_object_embeddings"], leaves["local_text_embeddings"], leaves["object_mask"], tlen, tmask)[0]
            loss.backward()
            return
        # only this part's parameters are touched (resetting every .grad detached the arena-backed views of the other tower)
        mods = [model.object_model] if part == "object" else [model.text_model, model.txt_proj]
        for mod in mods:
            for p in mod.parameters():
                p.grad = None
        if part == "object":
            emb, _ = model.object_model(obj, mask)
        else:
            g_, l_ = model.compute_text(data["text"])
            emb = torch.cat([g_.unsqueeze(1), l_], 1)
        if state["dy"] is None:
            state["dy"] = torch.randn(emb.shape, device=emb.device, dtype=torch.float32).mul_(1e-3).to(emb.dtype)
        emb.backward(state["dy"])
        Fn.join_side_stream()
        ops.flush_reductions()

    for _ in range(2):
        one()
    mode, g = "eager", None
    if graph:
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with ops.graph_capture(g):
                one()
            g.replay()
            torch.cuda.synchronize()
            mode = "hipGraph replay"
        except Exception as e:  # noqa: BLE001
            print("object tower: graph capture failed (%s), timing eager launches" % str(e)[:200], file=sys.stderr)
            g = None
            torch.cuda.synchronize()
    dist_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        if g is not None:
            g.replay()
        else:
            one()
    dist_sync()
    dt_ = (time.perf_counter() - t0) / steps
    # hand the arena-backed gradient views back (trainer.ParamArena: p.grad is a view of the flat gradient arena the optimizer and the
    # graph step read): whatever runs after this breakdown sees the parameters as the stepper left them
    for p in model.parameters():
        v = getattr(p, "_dvlp_grad_view", None)
        if v is not None:
            p.grad = v
    return dt_, mode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--regions", type=int, default=36)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap-wgrad", type=int, default=0,
                    help="side-stream gradient work: 0 off (default: per-kernel timings stay well defined), 1 bias-gradient reductions, "
                         "2 also weight-gradient GEMMs (+6%% pairs/s, but concurrent GEMMs stretch each other)")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=INT",
                    help="developer A/B switch: call the C-ABI setter NAME (e.g. dvlp_dev_xattn_gram=0) before the first step; repeatable")
    ap.add_argument("--p8", type=int, default=-1, help="override dvlp_dev_gemm_p8_mode (0 never / 1 heuristic / 2 always use the 256x256 GEMM kernel)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-launch HIP-event timing of the GEMMs")
    ap.add_argument("--no-object-tower", action="store_true", help="skip the object-transformer-only fwd+bwd timing")
    ap.add_argument("--parallel-towers", type=int, default=1, help="1 (default): text tower on its own HIP stream, concurrent with the object tower; 0: one stream")
    ap.add_argument("--text-dropout", type=float, default=-1.0,
                    help="DistilBERT dropout / attention_dropout probability; default: the HF config's 0.1, active in train mode as in the "
                         "reference (model/model.py:29-30); 0 switches it off")
    ap.add_argument("--graph", type=int, default=1,
                    help="1 (default): the step runs as one captured hipGraph (GraphedTrainStep; gradient all-reduce outside the graph "
                         "when N > 1); 0: eager launches with the hook-driven, backward-overlapped GradReducer")
    ap.add_argument("--rccl-channels", type=int, default=0, help="N > 1: cap RCCL's channel count (NCCL_MAX_NCHANNELS); 0 = RCCL's default")
    ap.add_argument("--exchange", default="all_reduce", choices=["all_reduce", "rs_ag"],
                    help="gradient exchange of the graph step: RCCL all_reduce per bucket (default) or reduce_scatter + all_gather per bucket (the direct, "
                         "all-links form: DESIGN section 5; unmeasured on > 1 GPU)")
    ap.add_argument("--grad-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="N > 1, graph mode: dtype of the gradient buckets on the links (bf16 = half the bytes; moments and master weights stay fp32)")
    ap.add_argument("--gather-negatives", action="store_true",
                    help="cross-GPU negatives: all-gather the embeddings for the contrastive losses (opt-in; the reference trains "
                         "with per-rank negatives, trainer/trainer_dist.py:148-165)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))          # before anything initialises the GPU in this process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist
    # developer / test switch: DVLP_BENCH_ONE_GPU=1 puts every rank on cuda:0 with gloo collectives, so the self-launch and the whole
    # N > 1 code path (graph + post-graph all-reduce, per-rank gathers, max-over-ranks timing) can be exercised on a one-GPU box
    one_gpu = os.environ.get("DVLP_BENCH_ONE_GPU") is not None
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.rccl_channels > 0:
        # RCCL beside CU-owning GEMMs: a channel is a workgroup that keeps a CU for the length of a collective; the cap is an
        # experiment knob for the first multi-GPU run (default: RCCL's own choice)
        os.environ["NCCL_MAX_NCHANNELS"] = str(a.rccl_channels)
        os.environ["NCCL_MIN_NCHANNELS"] = str(min(4, a.rccl_channels))
    force_dist = world == 1 and os.environ.get("DVLP_FORCE_DIST") is not None and "MASTER_ADDR" in os.environ   # developer switch:
    if world > 1 or force_dist:                                                                              # RCCL path on one GPU
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from demovlp_amd import ops, synthetic as syn
    from demovlp_amd.loss import GlobalLocalLoss
    from demovlp_amd.model import ObjectRelation
    from demovlp_amd.trainer import FusedAdamW, GradReducer, GraphedTrainStep, ParamArena, train_step

    B, F, R = a.batch, a.frames, a.regions
    cdt = "bfloat16" if a.dtype == "bf16" else "float32"
    model = ObjectRelation({"model": "", "input_objects": False, "object_num": R, "num_frames": F, "time_module": None},
                           {"model": "", "pretrained": True, "input": "text", "two_outputs": True}, pretrained_init=False, compute_dtype=cdt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(F, R).items()})
    model.to(dev)
    if a.text_dropout >= 0.0:
        model.set_text_dropout(a.text_dropout)
    import demovlp_amd.functional as Fn
    if a.p8 >= 0 or a.knob:
        # developer switches exist only in the -DDVLP_DEV build of the library; the default run loads the product library and flips nothing
        from demovlp_amd import _lib
        _lib.use_dev_library()
    Fn.OVERLAP_WGRAD = int(a.overlap_wgrad)
    Fn.OVERLAP_TEXT_ONLY = bool(int(os.environ.get("DVLP_OVERLAP_TEXT_ONLY", "0")))
    if a.p8 >= 0:
        ops.call("dvlp_dev_gemm_p8_mode", a.p8)
    for kv in a.knob:
        name, val = kv.split("=")
        ops.call(name, int(val))
    arena = ParamArena(model, bf16_shadow=(a.dtype == "bf16"))
    opt = FusedAdamW(arena, lr=1e-5)
    loss_fn = GlobalLocalLoss(use_local=True, use_global=True, coef=1.0, focal_type="equal")
    gather = None
    if a.gather_negatives and (world > 1 or force_dist):
        gather = argparse.Namespace(world_size=world, rank=rank)
    use_graph = bool(a.graph)
    model.parallel_towers = bool(a.parallel_towers)
    reducer, stepper = None, None
    if use_graph:
        stepper = GraphedTrainStep(model, loss_fn, opt, warmup=2, always_reduce=force_dist, grad_dtype="bfloat16" if a.grad_dtype == "bf16" else "float32",
                                   exchange=a.exchange, gather_negatives=gather)
    elif world > 1 or force_dist:
        reducer = GradReducer(arena, bucket_mb=64.0, always_reduce=force_dist)

    obj, mask = syn.fast_region_batch(B, F, R, seed=7 + rank)
    ids, att = syn.caption_batch(B, first_sample=rank * B)
    data = {"text": {"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(att).to(dev)},
            "object": torch.from_numpy(obj).to(dev), "object_mask": torch.from_numpy(mask).to(dev)}

    def sync():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if stepper is not None:
            return stepper(data)[0]
        return train_step(model, loss_fn, opt, data, reducer, gather_negatives=gather)[0]

    if use_graph:
        a.warmup = max(a.warmup, 3)              # two eager steps + the capturing one must precede the timed region
    loss = None
    step1_loss = None
    for i in range(a.warmup):
        loss = step()
        if i == 0:
            step1_loss = float(loss.item())
    if stepper is not None and stepper.inputs is not None:
        # the batch lives in the graphs' input buffers from here on -- where RegionBatcher.to_device(out=stepper.inputs) puts the
        # selection kernel's output in a real run -- so a step carries no device-to-device copy of its input
        ins = stepper.inputs
        for k, v in data["text"].items():
            ins["text"][k].copy_(v)
        ins["object"].copy_(data["object"])
        ins["object_mask"].copy_(data["object_mask"])
        data = ins
    sync()
    timing_inline = not a.no_kernel_timing and not use_graph
    if timing_inline:
        model.parallel_towers = False                    # per-launch event timings need one stream
        ops.XATTN_ONE_STREAM = True                      # (per-call option of dvlp_xattn_fwd / _bwd: no process-global switch is flipped)
        ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    host_elapsed = time.perf_counter() - t0          # time for the host to ENQUEUE the steps (launch-bound if ~= elapsed)
    sync()
    elapsed = time.perf_counter() - t0
    final_loss = float(loss.item())
    my_elapsed = elapsed
    eager_ms = None
    if not a.no_kernel_timing and use_graph:
        # per-launch HIP events cannot be recorded inside a captured graph: the GEMM family is timed over an equally long EAGER
        # region right behind the replayed one (same kernels, same shapes, same launch order, the stream they are launched on)
        model.parallel_towers = False                # one stream: per-launch event timings are only meaningful without concurrent kernels
        ops.XATTN_ONE_STREAM = True                  # (the local loss' two halves too -- a per-call option, DVLP_XATTN_ONE_STREAM: beside each other their products stretch one another)
        stepper._eager(data)
        sync()
        ops.prof_enable(True)
        t1 = time.perf_counter()
        for _ in range(a.steps):
            stepper._eager(data)
        sync()
        eager_ms = 1e3 * (time.perf_counter() - t1) / a.steps
    ops.XATTN_ONE_STREAM = False
    ops.prof_enable(False)
    gemm_ms, gemm_flops, gemm_n = ops.prof_collect() if not a.no_kernel_timing else (0.0, 0.0, 0)

    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # after the timed region: per-rank rates, a stand-alone all-reduce of the whole gradient arena, the object tower alone
    per_rank = None
    allreduce_ms = None
    if world > 1 or force_dist:
        rates = torch.zeros(world, device=dev, dtype=torch.float64)
        rates[rank] = B * a.steps / my_elapsed
        dist.all_reduce(rates)
        per_rank = [round(float(x), 1) for x in rates.tolist()]
        for it in range(4):
            if it == 1:
                sync()
                t1 = time.perf_counter()
            dist.all_reduce(arena.flat_g)
        sync()
        allreduce_ms = round(1e3 * (time.perf_counter() - t1) / 3, 3)
    exchange = None
    if (world > 1 or force_dist) and use_graph and getattr(stepper, "collective", False):
        # per-piece exchange times: three more steps with HIP events around every piece's collectives on the communication stream (outside the
        # timed region: the events make the host wait for each piece's handles)
        stepper.time_exchange = True
        for _ in range(3):
            stepper(data)
        xt = stepper.exchange_times()
        stepper.time_exchange = False
        npieces = len(stepper.piece_runs)
        exchange = []
        for k in range(npieces):                              # grouped by the piece index each entry carries: a piece with no ranges records none
            mine = [(t, n) for t, n, piece in xt if piece == k]
            exchange.append({"piece": k, "mb": round(mine[0][1] / 2 ** 20, 1) if mine else 0.0,
                             "ms": round(sum(t for t, _ in mine) / len(mine), 3) if mine else 0.0})
    obj_s, obj_mode, text_s, loss_s = None, None, None, None
    if not a.no_object_tower:
        obj_s, obj_mode = time_object_tower(model, data, max(3, min(a.steps, 10)), sync, graph=use_graph)
        try:          # the step's other two terms, timed the same way (breakdown only: a failure here must not cost the bench line)
            text_s = time_object_tower(model, data, max(3, min(a.steps, 10)), sync, graph=use_graph, part="text")[0]
            loss_s = time_object_tower(model, data, max(3, min(a.steps, 10)), sync, graph=use_graph, part="loss")[0]
        except Exception as e:  # noqa: BLE001
            print("tower breakdown failed: %s" % str(e)[:300], file=sys.stderr)

    if rank == 0:
        pairs = B * world * a.steps
        value = pairs / elapsed
        fpp, fpp_obj = flops_per_pair(B, F, R)
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        out = {
            "metric": "video-text pairs/sec/node (fwd+bwd+optimizer step) on 8-frame x 36-region synthetic",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "configs/pt/o2t-cl-local-select-loss-cc.json arch+loss, F=%d frames x R=%d regions x 2048-d synthetic "
                                   "region features + random 100-token captions, per-GPU batch %d" % (F, R, B),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "optimizer": "fused HF-AdamW",
                       "negatives": "all-gathered" if gather is not None else "per-rank (reference default)",
                       "text_dropout": model.text_model.config.dropout,
                       "step1_loss": None if step1_loss is None else round(step1_loss, 4), "final_loss": round(final_loss, 4)},
            "launch_mode": ("hipGraph replay (1 graph per step)" if use_graph else "eager") + (", text tower on its own stream" if a.parallel_towers else ""),
            "inputs": ("resident in HBM in the graphs' input buffers (where RegionBatcher.to_device(out=step.inputs) writes the selection kernel's output): no per-step copy"
                       if use_graph else "resident in HBM"),
            "host_enqueue_ms_per_step": round(1e3 * host_elapsed / a.steps, 3),
            "step_model_tflops": round(value * fpp / 1e12, 2),
            "step_frac_of_mfma_peak": round(value * fpp / 1e12 / (peak * world), 4),
        }
        if getattr(ops, "LN_FWD_ABLATE", False):
            out["INVALID_timing_only_ablation"] = "DVLP_ABLATE_LN_FWD=1: forward LayerNorm passes skipped (results wrong): an upper bound for LayerNorm fusion, not a throughput figure"
        if per_rank is not None:
            out["per_rank_pairs_per_s"] = per_rank
            out["grad_allreduce_ms_standalone"] = allreduce_ms
            out["grad_allreduce_bytes"] = int(arena.flat_g.numel() * 4)
            if exchange is not None:
                out["grad_exchange_pieces"] = exchange          # time each piece's collectives occupy the communication stream (all but the last run beside the next graph)
                out["grad_exchange_dtype"] = a.grad_dtype
                out["grad_exchange_form"] = {"requested": a.exchange, "used": sorted(getattr(stepper, "exchange_used", None) or [a.exchange])}
            if use_graph and getattr(stepper, "graph2", None) is not None:
                sizes = [sum(hi - lo for lo, hi in runs) * 4 >> 20 for runs in stepper.piece_runs]
                out["grad_exchange"] = ("backward captured as %d graphs cut at object blocks %s: the gradients a piece finishes (%s MB: text tower + top blocks first, "
                                        "lowest blocks + prologue + vector tail last) are all-reduced in %d MB buckets on a communication stream while the next graph "
                                        "runs, and each reduced bucket goes straight to the fused optimizer on a third stream; only the last piece's exchange is exposed"
                                        % (len(stepper.graphs), list(stepper.cuts), " / ".join(map(str, sizes)), stepper.bucket * 4 >> 20))
                out["launch_mode"] = out["launch_mode"].replace("1 graph per step", "%d graphs per step" % len(stepper.graphs))
        if gemm_n:
            # HBM-side traffic per launch of the GEMM family cannot be self-measured from inside the process: it comes from
            # the committed rocprofv3 PMC passes of this same command, regenerated every round (tools/profile_round.sh)
            traffic, traffic_src, traffic_note = None, None, None
            for fn in TRAFFIC_FILES:
                try:
                    tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
                except Exception:
                    continue
                if not (a.dtype == "bf16" and B == 64 and F == 8 and R == 36):
                    traffic_note = "the committed PMC passes are of the default workload (bf16, B=64, F=8, R=36)"
                elif tj.get("csrc_sha") != csrc_sha():
                    traffic_note = ("null: profiles/%s was measured on kernel sources %s (git %s), this tree's demovlp_amd/csrc hashes to %s -- re-run "
                                    "tools/profile_round.sh" % (fn, tj.get("csrc_sha"), tj.get("git_head"), csrc_sha()))
                else:
                    traffic = round(tj["gemm_family_traffic_bytes_per_launch"])
                    traffic_src = "profiles/%s (git %s, csrc %s)" % (fn, tj.get("git_head"), tj.get("csrc_sha"))
                break
            ach = gemm_flops / (gemm_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "traffic": traffic, "traffic_unit": "bytes/launch (L2<->fabric, rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, offline PMC pass)",
                               "traffic_source": traffic_src, **({"traffic_note": traffic_note} if traffic_note else {}),
                               "algorithmic_flops_per_launch": round(gemm_flops / gemm_n), "kernel": ("gemm_bf16_p8_kernel / gemm_bf16_p8_group_kernel / gemm_bf16_glds_kernel (all operand forms, incl. their split-K reductions)"
                                          if a.dtype == "bf16" else "gemm_f32_kernel (all forms)"),
                               "launches_per_step": gemm_n // a.steps, "avg_launch_us": round(1e3 * gemm_ms / gemm_n, 2),
                               # share of the region the launches were timed in (the eager pass behind a graph-replayed timed region)
                               "gemm_share_of_step": round(gemm_ms * 1e-3 / (eager_ms * 1e-3 * a.steps if eager_ms is not None else my_elapsed), 3),
                               "timed_over": ("%d eager steps right behind the %d graph-replayed ones (%.3f ms/step eager)" % (a.steps, a.steps, eager_ms))
                                             if eager_ms is not None else "the timed region itself"}
            if obj_s is not None:
                # north_star's target quantity: ObjectTransformer forward + backward alone, every kernel of it included
                out["roofline"]["object_transformer_ms"] = round(1e3 * obj_s, 3)
                out["roofline"]["object_transformer_launch_mode"] = obj_mode
                out["roofline"]["object_transformer_tflops"] = round(B * fpp_obj / obj_s / 1e12, 2)
                out["roofline"]["object_transformer_frac"] = round(B * fpp_obj / obj_s / 1e12 / peak, 4)
                # which figure is which: `frac` above is the GEMM family's launches alone, event-timed over an eager, one-stream pass (the only way
                # to bracket single launches); `object_transformer_frac` is the whole tower -- every kernel and every boundary of it -- timed on the
                # schedule the step itself runs on (hipGraph replay), and is the quantity north_star's >= 0.40 bar is stated on
                out["roofline"]["north_star_bar"] = {"applies_to": "object_transformer_frac", "target": 0.40}
            if text_s is not None and loss_s is not None:
                # the step's other terms alone, same launch mode: text tower (26.15 GFLOP per pair) and sim_matrix + GlobalLocalLoss forward + backward
                out["roofline"]["text_tower_ms"] = round(1e3 * text_s, 3)
                out["roofline"]["text_tower_frac"] = round(B * text_flops_per_pair(int(data["text"]["input_ids"].shape[1])) / text_s / 1e12 / peak, 4)
                out["roofline"]["loss_heads_ms"] = round(1e3 * loss_s, 3)
        if not gemm_n and obj_s is not None:        # (--no-kernel-timing: no roofline object; the object tower's figure stands alone)
            out["object_transformer"] = {"ms": round(1e3 * obj_s, 3), "launch_mode": obj_mode, "tflops": round(B * fpp_obj / obj_s / 1e12, 2),
                                         "frac": round(B * fpp_obj / obj_s / 1e12 / peak, 4)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(F, R)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
