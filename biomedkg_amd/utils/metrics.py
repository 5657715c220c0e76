"""Evaluation metrics of the KGE module (reference ``biomedkg/utils/metrics.py`` + the torchmetrics
classes ``kge_module.py:63-74`` instantiates).  torchmetrics is not installable here, so the few
classes the reference touches are restated with the same update/compute/reset surface.  These are
eval-side, outside the timed training step.
"""
from __future__ import annotations

import copy

import torch


class Metric(torch.nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self._defaults = {}

    def add_state(self, name, default, dist_reduce_fx=None):
        self._defaults[name] = default.clone()
        self.register_buffer(name, default.clone(), persistent=False)

    def reset(self):
        for k, v in self._defaults.items():
            getattr(self, k).copy_(v.to(getattr(self, k).device))


class _Binary(Metric):
    """Collects (preds, target); logits outside [0,1] are squashed with a sigmoid (torchmetrics rule)."""

    def __init__(self, task="binary", **kwargs):
        super().__init__()
        self.preds, self.target = [], []

    def update(self, preds, target):
        self.preds.append(preds.detach().float())
        self.target.append(target.detach())

    def reset(self):
        self.preds, self.target = [], []

    def _cat(self):
        p, t = torch.cat(self.preds), torch.cat(self.target)
        if ((p < 0) | (p > 1)).any():
            p = torch.sigmoid(p)
        return p, t.long()

    def compute(self):
        return self._compute(*self._cat())


class AUROC(_Binary):
    @staticmethod
    def _compute(p, t):
        n_pos, n_neg = int(t.sum()), int((1 - t).sum())
        if n_pos == 0 or n_neg == 0:
            return torch.tensor(0.0, device=p.device)
        order = torch.argsort(p)
        ps, ts = p[order], t[order]
        ranks = torch.arange(1, p.numel() + 1, device=p.device, dtype=torch.float64)
        uniq, inv, cnt = torch.unique_consecutive(ps, return_inverse=True, return_counts=True)  # mid-ranks for ties
        ends = torch.cumsum(cnt, 0).double()
        avg = ends - (cnt.double() - 1) / 2
        ranks = avg[inv]
        return ((ranks[ts == 1].sum() - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg)).float()


class AveragePrecision(_Binary):
    @staticmethod
    def _compute(p, t):
        if int(t.sum()) == 0:
            return torch.tensor(0.0, device=p.device)
        order = torch.argsort(p, descending=True)
        ps, ts = p[order], t[order].double()
        tp = torch.cumsum(ts, 0)
        last = torch.ones_like(ps, dtype=torch.bool)       # evaluate at the end of each tie group
        last[:-1] = ps[1:] != ps[:-1]
        prec = (tp / torch.arange(1, p.numel() + 1, device=p.device))[last]
        rec = (tp / ts.sum())[last]
        rec_prev = torch.cat([rec.new_zeros(1), rec[:-1]])
        return ((rec - rec_prev) * prec).sum().float()


class F1Score(_Binary):
    @staticmethod
    def _compute(p, t):
        pred = (p > 0.5).long()
        tp = ((pred == 1) & (t == 1)).sum().double()
        fp = ((pred == 1) & (t == 0)).sum().double()
        fn = ((pred == 0) & (t == 1)).sum().double()
        den = 2 * tp + fp + fn
        return (2 * tp / den).float() if den > 0 else torch.tensor(0.0, device=p.device)


class BootStrapper(Metric):
    """torchmetrics BootStrapper defaults: 10 bootstraps, poisson resampling, returns mean and std.
    ``poisson`` (class attribute, tests only): ``f(n, device, bootstrap_index) -> int64 [n]`` multiplicities in place of
    the device draw, so that a run can be replayed into the CPU checker."""

    poisson = None

    def __init__(self, base_metric, num_bootstraps: int = 10, **kwargs):
        super().__init__()
        self.metrics = torch.nn.ModuleList(copy.deepcopy(base_metric) for _ in range(num_bootstraps))

    def update(self, preds, target):
        n = preds.numel()
        for b, m in enumerate(self.metrics):
            cnt = (self.poisson(n, preds.device, b) if self.poisson is not None
                   else torch.poisson(torch.ones(n, device=preds.device)).long())
            idx = torch.repeat_interleave(torch.arange(n, device=preds.device), cnt)
            if idx.numel():
                m.update(preds[idx], target[idx])

    def compute(self):
        vals = torch.stack([m.compute() for m in self.metrics])
        return {"mean": vals.mean(), "std": vals.std()}

    def reset(self):
        for m in self.metrics:
            m.reset()


class MetricCollection(torch.nn.ModuleDict):
    def __init__(self, metrics, prefix=""):
        super().__init__()
        torch.nn.ModuleDict.update(self, metrics)   # `update` below is the METRIC update, not dict.update
        self.prefix = prefix

    def clone(self, prefix=""):
        c = copy.deepcopy(self)
        c.prefix = prefix
        return c

    def update(self, *args):
        for m in self.values():
            m.update(*args)

    def compute(self):
        out = {}
        for k, m in self.items():
            v = m.compute()
            if isinstance(v, dict):
                out.update({f"{self.prefix}{k}_{kk}": vv for kk, vv in v.items()})
            else:
                out[self.prefix + k] = v
        return out

    def reset(self):
        for m in self.values():
            m.reset()


class EdgeWisePrecision(Metric):
    """reference metrics.py:5-39: per relation, the fraction of positive logits above ``threshold``."""

    def __init__(self, class_mapping: dict, threshold: float = 0.5, **kwargs):
        super().__init__(**kwargs)
        self.threshold = threshold
        self.class_mapping = class_mapping
        self.num_classes = len(class_mapping)
        self.add_state("class_counts", default=torch.zeros(self.num_classes), dist_reduce_fx="sum")
        self.add_state("above_threshold_counts", default=torch.zeros(self.num_classes), dist_reduce_fx="sum")

    def update(self, preds, target):
        if self.class_counts.device != preds.device:
            self.to(preds.device)
        k = self.num_classes
        valid = (target >= 0) & (target < k)
        tgt = target[valid].long()
        self.class_counts += torch.bincount(tgt, minlength=k).float()
        self.above_threshold_counts += torch.bincount(tgt[preds[valid] > self.threshold], minlength=k).float()

    def compute(self) -> dict:
        counts, above = self.class_counts.tolist(), self.above_threshold_counts.tolist()
        return {self.class_mapping[c] + "_pre": (above[c] / counts[c] if counts[c] > 0 else 0.0)
                for c in range(self.num_classes)}
