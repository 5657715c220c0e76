"""CPU fp64 restatement of the torchmetrics classes ``KGEModule`` builds (reference ``biomedkg/kge_module.py:63-74``:
``MetricCollection{AUROC, AveragePrecision, F1Score}(task="binary")`` each inside a ``BootStrapper``) and of the
reference's own ``EdgeWisePrecision`` (``biomedkg/utils/metrics.py:5-39``).  TEST INFRASTRUCTURE (see oracle/__init__.py).

torchmetrics is an unpinned third-party dependency (pulled in by lightning==2.2.3, ``pyproject.toml:7``; NOT under
``/root/reference`` and not installable here), so this follows its published binary definitions:

* format: predictions outside [0, 1] are logits and go through a sigmoid (``_binary_*_format``);
* ``_binary_clf_curve``: sort by descending score, one threshold per DISTINCT score, ``tps = cumsum(target)`` and
  ``fps = 1 + idx - tps`` at the last index of every tie group;
* AUROC = trapezoidal area under (fpr, tpr) with the origin prepended; 0 when a class is absent;
* AveragePrecision = ``-sum((recall[1:] - recall[:-1]) * precision[:-1])`` over the reversed curve closed with (1, 0);
* F1 = ``2 tp / (2 tp + fp + fn)`` at threshold 0.5 (``preds > 0.5``), 0 when the denominator is 0;
* BootStrapper(num_bootstraps=10, sampling_strategy="poisson", mean=True, std=True): every update resamples its batch
  with multiplicities ``n_i ~ Poisson(1)`` per bootstrap (``repeat_interleave``); compute = mean / unbiased std of
  the bootstraps' values.
``tests/test_oracle_known_answers.py`` pins the three scores against scikit-learn (an independent implementation).
"""
from __future__ import annotations

import numpy as np


def _format(preds, target):
    p = np.asarray(preds).reshape(-1)
    t = np.asarray(target).reshape(-1).astype(np.int64)
    if ((p < 0) | (p > 1)).any():
        # torchmetrics squashes in the predictions' own dtype: fp32 logits saturate (and tie) where fp64 would not
        p = (1.0 / (1.0 + np.exp(-p))).astype(p.dtype) if p.dtype == np.float32 else 1.0 / (1.0 + np.exp(-p.astype(np.float64)))
    return p.astype(np.float64), t


def _clf_curve(p, t):
    order = np.argsort(-p, kind="stable")
    p, t = p[order], t[order]
    last = np.r_[np.nonzero(p[1:] != p[:-1])[0], p.size - 1]       # last index of every tie group
    tps = np.cumsum(t)[last].astype(np.float64)
    fps = 1.0 + last - tps
    return fps, tps


def binary_auroc(preds, target):
    p, t = _format(preds, target)
    if t.sum() == 0 or (1 - t).sum() == 0:
        return 0.0
    fps, tps = _clf_curve(p, t)
    fpr, tpr = np.r_[0.0, fps] / fps[-1], np.r_[0.0, tps] / tps[-1]
    return float(np.sum((fpr[1:] - fpr[:-1]) * (tpr[1:] + tpr[:-1]) / 2.0))


def binary_average_precision(preds, target):
    p, t = _format(preds, target)
    if t.sum() == 0:
        return 0.0
    fps, tps = _clf_curve(p, t)
    precision = np.r_[(tps / (tps + fps))[::-1], 1.0]
    recall = np.r_[(tps / tps[-1])[::-1], 0.0]
    return float(-np.sum((recall[1:] - recall[:-1]) * precision[:-1]))


def binary_f1(preds, target, threshold=0.5):
    p, t = _format(preds, target)
    pred = p > threshold
    tp, fp, fn = np.sum(pred & (t == 1)), np.sum(pred & (t == 0)), np.sum(~pred & (t == 1))
    den = 2.0 * tp + fp + fn
    return float(2.0 * tp / den) if den > 0 else 0.0


def bootstrap(fn, batches, counts):
    """``batches``: [(preds, target), ...] in update order; ``counts[b][u]``: the Poisson multiplicities bootstrap b
    drew for update u.  -> (mean, unbiased std) of ``fn`` over the bootstraps."""
    vals = []
    for per_update in counts:
        ps, ts = [], []
        for (p, t), c in zip(batches, per_update):
            idx = np.repeat(np.arange(len(c)), np.asarray(c, dtype=np.int64))
            if idx.size:
                ps.append(np.asarray(p)[idx]); ts.append(np.asarray(t)[idx])     # keeps the dtype: see _format
        vals.append(fn(np.concatenate(ps), np.concatenate(ts)))
    vals = np.asarray(vals, dtype=np.float64)
    return float(vals.mean()), float(vals.std(ddof=1))


def edge_wise_precision(pos_preds, edge_type, class_mapping, threshold=0.5):
    """reference metrics.py:20-39: per relation, the share of the positives' RAW logits above ``threshold``."""
    p, t = np.asarray(pos_preds, dtype=np.float64), np.asarray(edge_type)
    out = {}
    for c in range(len(class_mapping)):
        m = t == c
        out[class_mapping[c] + "_pre"] = float((p[m] > threshold).sum() / m.sum()) if m.sum() > 0 else 0.0
    return out
