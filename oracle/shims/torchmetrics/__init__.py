"""Shim torchmetrics: state containers only; metric values are not part of the goldens."""
import copy

import torch


class Metric(torch.nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self._defaults = {}

    def add_state(self, name, default, dist_reduce_fx=None):
        self._defaults[name] = default.clone()
        setattr(self, name, default.clone())

    def reset(self):
        for k, v in self._defaults.items():
            setattr(self, k, v.clone())


class _Collect(Metric):
    def __init__(self, task=None, **kwargs):
        super().__init__()
        self.preds, self.target = [], []

    def update(self, preds, target):
        self.preds.append(preds.detach())
        self.target.append(target.detach())

    def compute(self):
        return torch.tensor(float("nan"))

    def reset(self):
        self.preds, self.target = [], []


class AUROC(_Collect):
    pass


class AveragePrecision(_Collect):
    pass


class F1Score(_Collect):
    pass


class MetricCollection(torch.nn.ModuleDict):
    def __init__(self, metrics, prefix=""):
        super().__init__(metrics)
        self.prefix = prefix

    def clone(self, prefix=""):
        c = copy.deepcopy(self)
        c.prefix = prefix
        return c

    def update(self, *args):
        for m in self.values():
            m.update(*args)

    def compute(self):
        return {self.prefix + k: m.compute() for k, m in self.items()}

    def reset(self):
        for m in self.values():
            m.reset()
