"""Round-4 parity coverage on a real MI355X:

* retrieval evaluation where retrieval WORKS (golden G11: t2v R@1 91 %, v2t R@1 23 %, MedR 8 on 256 pairs; G9 sits at chance): the fp32
  path must reproduce every query's rank -- hence R@1/5/10/50, MedR, MeanR -- except for queries whose decision the reference itself
  makes by less than the 1e-4 similarity bar; the bf16 path within a stated number of rank changes
  (trainer/trainer_dist.py:358-399, model/metric.py:10-122);
* bf16 training fidelity at the benchmark size: 20 graph-replayed optimisation steps at B = 64, F = 8, R = 36 against the fp32 HIP
  path on the same batches, at the config's lr 1e-5 and at the 2e-4 the reference's schedule quirk switches to
  (trainer/trainer_dist.py:144-171): per-step loss deviation, direction and size of the parameter update.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from demovlp_amd import synthetic as syn  # noqa: E402
from demovlp_amd.loss import GlobalLocalLoss  # noqa: E402
from demovlp_amd.model import ObjectRelation  # noqa: E402
from demovlp_amd.trainer import evaluate  # noqa: E402
from helpers import load_golden, rel_err  # noqa: E402

DEV = "cuda"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _retrieval_model(F, R, sd, dtype):
    m = ObjectRelation({"model": "", "input_objects": False, "object_num": R, "num_frames": F, "time_module": None},
                       {"model": "pretrained/distilbert-base-uncased", "pretrained": True, "input": "text", "two_outputs": True}, pretrained_init=False, compute_dtype=dtype)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.set_text_dropout(0.0)
    return m.to(DEV)


def _retrieval_batches(sd, F, R, BS, NB):
    for b in range(NB):
        obj, mask, ids, att = syn.retrieval_batch(sd, F, R, b * BS, BS)
        yield {"text": {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(att).to(DEV)},
               "object": torch.from_numpy(obj).to(DEV), "object_mask": torch.from_numpy(mask).to(DEV)}


def _match_ranks(sims, axis):
    """Rank (0 = best) of the matching item for every query: t2v queries are rows (axis 1 runs over videos), v2t queries columns."""
    s = sims if axis == 1 else sims.T
    d = np.diag(s)
    return (s > d[:, None]).sum(1)


def _ambiguous(sims, axis, tol):
    """Queries whose match competes with another item inside +-tol: the only ones whose rank a deviation below tol / 2 can change."""
    s = sims if axis == 1 else sims.T
    d = np.diag(s)
    close = np.abs(s - d[:, None]) < tol
    np.fill_diagonal(close, False)
    return close.any(1)


KEYS = ("R1", "R5", "R10", "R50", "MedR", "MeanR")


def test_fp32_evaluate_reproduces_every_rank_of_the_retrieval_set_with_signal():
    g = load_golden("g11_retrieval.npz")
    F, R, BS, NB = int(g["F"]), int(g["R"]), int(g["batch"]), int(g["batches"])
    n = BS * NB
    sd = syn.retrieval_state_dict(F, R)
    model = _retrieval_model(F, R, sd, "float32")
    res = evaluate(model, GlobalLocalLoss(use_local=True, use_global=True, coef=1.0, focal_type="equal"), _retrieval_batches(sd, F, R, BS, NB))
    assert abs(res["val_loss"] - g["val_losses"][:, 0].mean()) < 1e-4 * g["val_losses"][0, 0]
    assert rel_err(res["global_sims"], g["global_sims"]) < 1e-4 and rel_err(res["local_sims"], g["local_sims"]) < 1e-4
    dev = np.abs(res["o2t_sims"] - g["o2t_sims"]).max()
    assert dev < 1e-4 * max(1.0, np.abs(g["o2t_sims"]).max())
    for name, axis in (("t2v", 1), ("v2t", 0)):
        want, got = _match_ranks(g["o2t_sims"], axis), _match_ranks(res["o2t_sims"], axis)
        amb = _ambiguous(g["o2t_sims"], axis, 2.0 * dev)
        changed = want != got
        print("\n%s: max |sim dev| %.2e; %d of %d queries decided by less than twice that; ranks changed: %d" % (name, dev, amb.sum(), n, changed.sum()))
        assert not (changed & ~amb).any()                        # every clearly decided query keeps its exact rank
        assert np.abs(want - got).max() <= max(1, amb.sum())     # ... and an ambiguous one moves by the few neighbours it is tied with
        m = res["nested_val_metrics"][name + "_metrics"]
        ref = dict(zip(KEYS, g[name][:6]))
        if not changed.any():
            assert all(abs(m[k] - ref[k]) < 1e-9 for k in KEYS), (name, m, ref)          # R@1/5/10/50, MedR, MeanR: exact
        else:
            assert all(abs(m[k] - ref[k]) <= 100.0 * changed.sum() / n + 1e-9 for k in KEYS[:4]) and abs(m["MeanR"] - ref["MeanR"]) <= np.abs(want - got).sum() / n + 1e-9
    assert g["t2v"][0] > 50.0 and 10.0 < g["v2t"][0] < 60.0      # the set is neither at chance nor saturated


# bf16 bounds: measured on MI355X (printed by the test), then doubled
BF16_G11_SIM_TOL = 2e-2           # observed 7.6e-3
BF16_G11_R_AT_K = 4.0            # percentage points on R@1/5/10/50 (observed: 1.6, with 104 of 256 v2t ranks moved by up to 6 places)
BF16_G11_MEANR = 1.0             # observed 0.01


def test_bf16_evaluate_on_the_retrieval_set_with_signal_stays_within_stated_rank_changes():
    g = load_golden("g11_retrieval.npz")
    F, R, BS, NB = int(g["F"]), int(g["R"]), int(g["batch"]), int(g["batches"])
    n = BS * NB
    sd = syn.retrieval_state_dict(F, R)
    model = _retrieval_model(F, R, sd, "bfloat16")
    res = evaluate(model, GlobalLocalLoss(use_local=True, use_global=True, coef=1.0, focal_type="equal"), _retrieval_batches(sd, F, R, BS, NB))
    d = np.abs(res["o2t_sims"] - g["o2t_sims"]).max() / np.abs(g["o2t_sims"]).max()
    assert d < BF16_G11_SIM_TOL, d
    for name, axis in (("t2v", 1), ("v2t", 0)):
        want, got = _match_ranks(g["o2t_sims"], axis), _match_ranks(res["o2t_sims"], axis)
        m = res["nested_val_metrics"][name + "_metrics"]
        ref = dict(zip(KEYS, g[name][:6]))
        print("\n%s bf16: o2t rel dev %.2e; %d of %d ranks changed (max by %d); R@1/5/10/50 %s vs %s; MedR %s vs %s; MeanR %.2f vs %.2f"
              % (name, d, (want != got).sum(), n, np.abs(want - got).max(), [round(m[k], 2) for k in KEYS[:4]], np.round(g[name][:4], 2), m["MedR"], ref["MedR"],
                 m["MeanR"], ref["MeanR"]))
        assert all(abs(m[k] - ref[k]) <= BF16_G11_R_AT_K for k in KEYS[:4]), (name, m, ref)
        assert abs(m["MedR"] - ref["MedR"]) <= 2.0 and abs(m["MeanR"] - ref["MeanR"]) <= BF16_G11_MEANR
        assert m["R1"] > 0.5 * ref["R1"]                                                  # the signal survives bf16


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 4 end to end (golden G14): fine-tune, THEN retrieve -- trainer/trainer_dist.py:104-203 followed by :205-408
# ---------------------------------------------------------------------------------------------------------------------
def _finetune_then_eval(dtype, lr, g):
    from demovlp_amd.trainer import FusedAdamW, ParamArena, train_step
    F, R, BS, NB, STEPS, FIRST = int(g["F"]), int(g["R"]), int(g["batch"]), int(g["batches"]), int(g["steps"]), int(g["first_train_pair"])
    sd = syn.retrieval_state_dict(F, R)
    model = _retrieval_model(F, R, sd, dtype)
    model.train()
    arena = ParamArena(model, bf16_shadow=(dtype == "bfloat16"))
    opt = FusedAdamW(arena, lr=lr)
    lf = GlobalLocalLoss(use_local=True, use_global=True, coef=1.0, focal_type="equal")
    curve = []
    for step in range(STEPS):
        obj, mask, ids, att = syn.retrieval_batch(sd, F, R, FIRST + step * BS, BS)
        data = {"text": {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(att).to(DEV)},
                "object": torch.from_numpy(obj).to(DEV), "object_mask": torch.from_numpy(mask).to(DEV)}
        curve.append([float(t.item()) for t in train_step(model, lf, opt, data)])
    model.eval()
    return np.array(curve), evaluate(model, lf, _retrieval_batches(sd, F, R, BS, NB))


# measured on MI355X (printed by the tests), bounds = 2-3x the observation
# lr 1e-5: curve 1.1e-5 of the loss, similarities 8.6e-4 (Adam turns fp32 rounding noise on near-zero gradients into +-lr moves: the weights
# after ten steps agree to ~1e-5, the similarity matrix built from them to 1e-3), 0 / 10 of 256 ranks moved (all among the queries the reference
# decides by less than twice that).  lr 2e-4 is CHAOTIC on this set: the reference's own curve climbs 15.84 -> 16.08 over its last five steps, two
# fp32 implementations part by 7e-3 of the loss after ten steps and their similarity matrices by 0.5 -- only the first steps are comparable.
G14_FP32 = {"lr1e-5": dict(curve=1e-4, sims=4e-3), "lr2e-4": dict(curve=3e-2, sims=None, first=(4, 2e-3))}


@pytest.mark.parametrize("tag,lr", [("lr1e-5", 1e-5), ("lr2e-4", 2e-4)])
def test_fp32_finetune_then_evaluate_vs_reference(tag, lr):
    """Golden G14: the imported reference fine-tunes 10 steps at the MSRVTT fine-tune geometry (F = 8, R = 30, B = 32, HF-AdamW) from the
    retrieval weights and then validates on the 256-pair set.  The fp32 HIP path, driven the same way (train_step x 10, evaluate), must
    follow the loss curve, land on the same similarity matrix up to the optimisation's own rounding sensitivity, keep the exact rank of
    every query the reference decides by more than twice that deviation, and so reproduce R@1/5/10/50, MedR, MeanR."""
    g = load_golden("g14_finetune_eval.npz")
    curve, res = _finetune_then_eval("float32", lr, g)
    ref = g[tag + "_curve"]
    dc = np.abs(curve - ref).max() / np.abs(ref[:, 0]).max()
    sims = g[tag + "_o2t_sims"].astype(np.float64)
    dev = np.abs(res["o2t_sims"] - sims).max()
    n = sims.shape[0]
    print("\nG14 %s fp32: loss curve dev %.2e of the loss (first %.4f -> last %.4f; reference %.4f -> %.4f); o2t sims max dev %.2e (max |sim| %.2f); val loss %.4f vs %.4f"
          % (tag, dc, curve[0, 0], curve[-1, 0], ref[0, 0], ref[-1, 0], dev, np.abs(sims).max(), res["val_loss"], g[tag + "_val_losses"][:, 0].mean()))
    assert dc < G14_FP32[tag]["curve"], dc
    if "first" in G14_FP32[tag]:                       # the chaotic learning rate: the first steps must still agree closely
        k, tol = G14_FP32[tag]["first"]
        d0 = np.abs(curve[:k] - ref[:k]).max() / np.abs(ref[:k, 0]).max()
        print("   first %d steps: %.2e of the loss" % (k, d0))
        assert d0 < tol, d0
    if G14_FP32[tag]["sims"] is None:
        assert np.isfinite(res["o2t_sims"]).all() and abs(res["val_loss"] - g[tag + "_val_losses"][:, 0].mean()) < 0.05 * g[tag + "_val_losses"][0, 0]
        return
    assert dev < G14_FP32[tag]["sims"] * max(1.0, np.abs(sims).max()), dev
    assert abs(res["val_loss"] - g[tag + "_val_losses"][:, 0].mean()) < 10 * G14_FP32[tag]["curve"] * g[tag + "_val_losses"][0, 0]
    for name, axis in (("t2v", 1), ("v2t", 0)):
        want, got = _match_ranks(sims, axis), _match_ranks(res["o2t_sims"], axis)
        amb = _ambiguous(sims, axis, 2.0 * dev)
        changed = want != got
        m = res["nested_val_metrics"][name + "_metrics"]
        refm = dict(zip(KEYS, g[tag + "_" + name][:6]))
        print("   %s: %d of %d queries decided by less than twice the deviation; ranks changed: %d; R@1/5/10/50 %s vs %s, MedR %s vs %s, MeanR %.3f vs %.3f"
              % (name, amb.sum(), n, changed.sum(), [round(m[k], 2) for k in KEYS[:4]], np.round(g[tag + "_" + name][:4], 2), m["MedR"], refm["MedR"], m["MeanR"], refm["MeanR"]))
        assert not (changed & ~amb).any()                        # every clearly decided query keeps its exact rank
        if not changed.any():
            assert all(abs(m[k] - refm[k]) < 1e-9 for k in KEYS), (name, m, refm)
        else:
            assert all(abs(m[k] - refm[k]) <= 100.0 * changed.sum() / n + 1e-9 for k in KEYS[:4])
            assert abs(m["MeanR"] - refm["MeanR"]) <= np.abs(want - got).sum() / n + 1e-9
    # the fine-tune did something a metric can see (G11, before it: t2v R@1 91.0, v2t R@1 23.0)
    g11 = load_golden("g11_retrieval.npz")
    assert g[tag + "_t2v"][0] != g11["t2v"][0] or g[tag + "_v2t"][0] != g11["v2t"][0]


# measured on MI355X (printed by the test), bounds = 2x the observation: ten bf16 optimisation steps move the similarity matrix more than the
# bf16 forward alone does (G11: 7.6e-3)
BF16_G14_SIM_TOL = 9e-2          # observed 4.3e-2 of max |sim|
BF16_G14_R_AT_K = 4.0            # percentage points (observed 1.6: 2 / 100 of 256 t2v / v2t ranks moved, by at most 1 / 14 places)
BF16_G14_MEANR = 1.0             # observed 0.24


def test_bf16_finetune_then_evaluate_stays_within_the_stated_bounds():
    """The same through the bf16 MFMA path (bf16 weight shadows, fp32 masters and moments) at the config's lr: loss curve within 2e-3 of the
    reference's, retrieval metrics within the bf16 bounds of G11 (4 points on R@K, MedR within 2, MeanR within 1)."""
    g = load_golden("g14_finetune_eval.npz")
    curve, res = _finetune_then_eval("bfloat16", 1e-5, g)
    ref = g["lr1e-5_curve"]
    dc = np.abs(curve - ref).max() / np.abs(ref[:, 0]).max()
    sims = g["lr1e-5_o2t_sims"].astype(np.float64)
    d = np.abs(res["o2t_sims"] - sims).max() / np.abs(sims).max()
    print("\nG14 bf16: loss curve dev %.2e of the loss; o2t rel dev %.2e" % (dc, d))
    bad = []
    if not (dc < 2e-3 and d < BF16_G14_SIM_TOL):
        bad.append(("curve / sims", dc, d))
    for name, axis in (("t2v", 1), ("v2t", 0)):
        m = res["nested_val_metrics"][name + "_metrics"]
        refm = dict(zip(KEYS, g["lr1e-5_" + name][:6]))
        want, got = _match_ranks(sims, axis), _match_ranks(res["o2t_sims"], axis)
        print("   %s bf16: %d ranks changed (max by %d); R@1/5/10/50 %s vs %s; MedR %s vs %s; MeanR %.2f vs %.2f"
              % (name, (want != got).sum(), np.abs(want - got).max(), [round(m[k], 2) for k in KEYS[:4]], np.round(g["lr1e-5_" + name][:4], 2), m["MedR"], refm["MedR"], m["MeanR"], refm["MeanR"]))
        if not all(abs(m[k] - refm[k]) <= BF16_G14_R_AT_K for k in KEYS[:4]):
            bad.append((name, "R@K", [round(m[k] - refm[k], 2) for k in KEYS[:4]]))
        if not (abs(m["MedR"] - refm["MedR"]) <= 2.0 and abs(m["MeanR"] - refm["MeanR"]) <= BF16_G14_MEANR):
            bad.append((name, "MedR / MeanR", m["MedR"] - refm["MedR"], m["MeanR"] - refm["MeanR"]))
    assert not bad, bad


# measured on MI355X (profiles/r4_bf16_fidelity.txt), bounds = 2-2.5x the observation:
#   lr 1e-5, 20 steps: max |loss16 - loss32| 3.4e-3 (1.9e-4 of the loss), |p16 - p32| = 0.115 |p32 - p0|, cos(update16, update32) 0.9934
#   lr 2e-4, 10 steps: 4.6e-2 at the step-3 spike (2.4e-3 of the loss)
# (HIP bf16 against HIP fp32: a drift bound, not parity -- the reference-anchored check at this size is golden G12, tests/test_gpu_round5.py;
#  the lr 2e-4 variant runs with --runslow)
@pytest.mark.parametrize("lr,steps,rel_tol,drift_tol,cos_min", [(1e-5, 20, 5e-4, 0.25, 0.98), pytest.param(2e-4, 10, 6e-3, 0.6, 0.90, marks=pytest.mark.slow)],
                         ids=["lr1e-5", "lr2e-4"])
def test_bf16_trains_like_fp32_at_the_benchmark_size(lr, steps, rel_tol, drift_tol, cos_min):
    from bf16_fidelity import fidelity
    r = fidelity(lr, steps=steps, B=64, verbose=False)
    print("\nlr %g, %d steps at B = 64: max |loss16 - loss32| %.2e (%.2e of the loss); |p16 - p32| / |p32 - p0| = %.3f; cos(update16, update32) = %.4f; "
          "loss %.4f -> %.4f (fp32) / %.4f -> %.4f (bf16)" % (lr, steps, r["max_dev"], r["max_rel"], r["drift"], r["cos"], r["l32"][0, 0], r["l32"][-1, 0],
                                                            r["l16"][0, 0], r["l16"][-1, 0]))
    assert r["max_rel"] < rel_tol and r["drift"] < drift_tol and r["cos"] > cos_min
    assert np.isfinite(r["l16"]).all() and r["l16"][-1, 0] < r["l16"][0, 0]                # it descends, as the fp32 run does
    assert abs((r["l16"][0, 0] - r["l16"][-1, 0]) - (r["l32"][0, 0] - r["l32"][-1, 0])) < 0.1 * abs(r["l32"][0, 0] - r["l32"][-1, 0]) + 2 * r["max_dev"]


def test_text_mask_len_kernel_equals_the_stock_ops():
    """dvlp_text_mask_len (one launch inside the captured step) against the reference's own expressions, trainer/trainer_dist.py:152-159: exact."""
    from demovlp_amd import ops
    for B, L in ((64, 100), (3, 2), (5, 37), (1, 300)):
        ids, att = syn.caption_batch(B, text_len=max(L, 34))
        att = torch.from_numpy(att[:, :L].copy()).to(DEV)
        att[:, 0] = 1
        length, mask = ops.text_mask_len(att)
        assert length.dtype == torch.int64 and torch.equal(length, torch.sum(att, dim=1))
        assert mask.dtype == torch.float32 and torch.equal(mask, (att[:, 1:] - 1.0) * 100.0)
