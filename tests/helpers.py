"""Shared test helpers: golden loading, synthetic batches built through the ORACLE region-select."""
import os

import numpy as np
import torch

from demovlp_amd import synthetic as syn
from oracle import restatement as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def n_raw_for(sample):
    return (36, 28, 50, 33)[sample % 4]     # must match tests/golden/make_golden.py


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def oracle_clip(sample, F, R):
    frames = [syn.make_frame(sample, f, n_raw_for(sample)) for f in range(F)]
    obj, mask, lens, orders = orc.region_select([fr["x"] for fr in frames], [fr["bbox"] for fr in frames],
                                                [fr["objects_conf"] for fr in frames], 640, 360, R)
    return obj, mask, lens, orders


def golden_batch(F, R, B):
    """The exact batch make_golden.py fed to the reference, rebuilt from seeds (numpy)."""
    objs, masks = [], []
    for s in range(B):
        o, m, _, _ = oracle_clip(s, F, R)
        objs.append(o)
        masks.append(m)
    ids, att = syn.caption_batch(B)
    return np.stack(objs), np.stack(masks), ids, att


def eval_batch(F, R, B, first):
    """Batch `first // B` of the G9 eval set (make_golden.py:golden_eval): samples first .. first+B-1."""
    objs, masks = [], []
    for s in range(first, first + B):
        o, m, _, _ = oracle_clip(s, F, R)
        objs.append(o)
        masks.append(m)
    ids, att = syn.caption_batch(B, first_sample=first)
    return np.stack(objs), np.stack(masks), ids, att


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def eval_grid_inputs():
    """Inputs of the G7 eval-grid golden (tests/golden/make_golden.py:golden_metrics): 11 videos x 22 captions, G=72, W=99."""
    nv, nt, G, W = 11, 22, 72, 99
    rng = np.random.default_rng(77)
    im = rng.standard_normal((nv, G, 256), dtype=np.float32)
    cap = rng.standard_normal((nt, W, 256), dtype=np.float32)
    for b in range(nt):
        cap[b, :G, :64] += 0.5 * im[b // 2, :, :64]
    m_img = np.zeros((nv, G), np.float32)
    m_img[3, G - 6:] = -100.0
    lens = rng.integers(5, 30, nt)
    m_cap = np.full((nt, W), -100.0, np.float32)
    for b in range(nt):
        m_cap[b, : lens[b]] = 0.0
    return im, cap, m_img, lens, m_cap

