"""Retrieval metrics: host-side numpy restatement of model/metric.py:10-214, 298-308 (t2v_metrics, v2t_metrics,
cols2metrics).  In the reference this is CPU numpy as well (an N x N matrix after all-gather); it is here so the
R@1/5/10/50, MedR, MeanR yardstick of BASELINE config 4 travels with the package.  Ranks are computed by counting
instead of sorting + matching, which yields the same numbers (optimistic tie-break for t2v, averaged ties for v2t)."""
from __future__ import annotations

import numpy as np


def cols2metrics(cols, num_queries):
    cols = np.asarray(cols, dtype=np.float64)
    m = {
        "R1": 100 * float(np.sum(cols == 0)) / num_queries,
        "R5": 100 * float(np.sum(cols < 5)) / num_queries,
        "R10": 100 * float(np.sum(cols < 10)) / num_queries,
        "R50": 100 * float(np.sum(cols < 50)) / num_queries,
        "MedR": np.median(cols) + 1,
        "MeanR": np.mean(cols) + 1,
    }
    stats = np.array([m["R1"], m["R5"], m["R10"]], dtype=np.float64)
    with np.errstate(divide="ignore"):
        m["geometric_mean_R1-R5-R10"] = float(np.exp(np.mean(np.log(stats))))     # scipy.stats.mstats.gmean
    return m


def t2v_metrics(sims, query_masks=None):
    """sims [num_queries, num_vids], x_ij = <text_i, video_j>; query i's ground truth is video i // queries_per_video."""
    sims = np.asarray(sims)
    assert sims.ndim == 2, "expected a matrix"
    nq, nv = sims.shape
    dists = -sims
    qpv = nq // nv
    gt = dists[np.arange(nq), np.arange(nq) // qpv][:, None]
    cols = (dists < gt).sum(axis=1)                       # first sorted position equal to the GT distance
    if query_masks is not None:
        keep = np.asarray(query_masks).reshape(-1).astype(bool)
        assert keep.size == nq, "invalid query mask shape"
        cols = cols[keep]
        nq = int(keep.sum())
    return cols2metrics(cols, nq)


def v2t_metrics(sims, query_masks=None):
    """sims as for t2v (text x video); ranks the closest ground-truth caption of every video, ties averaged."""
    sims = np.asarray(sims).T
    assert sims.ndim == 2, "expected a matrix"
    nq, ncap = sims.shape
    dists = (-sims).astype(np.float64, copy=True)
    cpv = ncap // nq
    MISSING = 1e8
    if query_masks is not None:
        dists[:, np.logical_not(np.asarray(query_masks).reshape(-1).astype(bool))] = MISSING
    ranks = np.empty(nq)
    for ii in range(nq):
        row = dists[ii]
        best = np.inf
        for jj in range(ii * cpv, (ii + 1) * cpv):
            if row[jj] == MISSING:
                continue
            less = np.sum(row < row[jj])
            eq = np.sum(row == row[jj])
            best = min(best, less + (eq - 1) / 2.0)
        ranks[ii] = best
    return cols2metrics(ranks, nq)
