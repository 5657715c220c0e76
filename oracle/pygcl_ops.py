"""Restatement of the PyGCL 0.1.2 contrast models on the BioMedKG hot path (test infrastructure).

PyGCL is pinned by the reference (``/root/reference/pyproject.toml:8``); its source is not under
``/root/reference`` and it is not installable here.  Follows SURVEY.md Appendix A.5 / A.6.
PARITY UNPINNED (no reference test covers it); cross-checked against ``F.cross_entropy`` /
closed forms in ``tests/test_oracle_known_answers.py``.

Reference call sites: ``biomedkg/gcl_module.py:127`` (SingleBranchContrast(JSD, "G2L")),
``biomedkg/gcl_module.py:171-173`` (DualBranchContrast(InfoNCE(tau=0.2), "L2L", intraview_negs=True)).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def infonce_compute(anchor, sample, pos_mask, neg_mask, tau: float):
    """A.5 ``InfoNCE.compute``: no max-subtraction, materialised ``[N, M]``."""
    sim = F.normalize(anchor) @ F.normalize(sample).t() / tau
    exp_sim = torch.exp(sim) * (pos_mask + neg_mask)
    log_prob = sim - torch.log(exp_sim.sum(dim=1, keepdim=True))
    loss = log_prob * pos_mask
    loss = loss.sum(dim=1) / pos_mask.sum(dim=1)
    return -loss.mean()


def dual_branch_l2l_intraview(h1, h2, tau: float):
    """A.5 ``DualBranchContrast(mode="L2L", intraview_negs=True)``: ``(l(h1,h2)+l(h2,h1))/2``."""
    n = h1.size(0)
    eye = torch.eye(n, dtype=h1.dtype, device=h1.device)
    zeros = torch.zeros_like(eye)
    pos = torch.cat([eye, zeros], dim=1)
    neg = torch.cat([1.0 - eye, 1.0 - eye], dim=1)

    def one(anchor, sample):
        return infonce_compute(anchor, torch.cat([sample, anchor], dim=0), pos, neg, tau)

    return (one(h1, h2) + one(h2, h1)) * 0.5


def jsd_g2l(h, g, hn):
    """A.6 ``SingleBranchContrast(JSD(), "G2L")(h=z, g=summary, hn=zn)``."""
    n = h.size(0)
    sample = torch.cat([h, hn], dim=0)
    pos = torch.cat([torch.ones(1, n), torch.zeros(1, n)], dim=1).to(h)
    neg = 1.0 - pos
    sim = g @ sample.t()
    e_pos = (math.log(2.0) - F.softplus(-sim * pos)).sum() / pos.sum()
    neg_sim = sim * neg
    e_neg = (F.softplus(-neg_sim) + neg_sim - math.log(2.0)).sum() / neg.sum()
    return e_neg - e_pos


class InfoNCE:
    def __init__(self, tau):
        self.tau = tau


class JSD:
    pass


class DualBranchContrast(torch.nn.Module):
    def __init__(self, loss, mode, intraview_negs=False, **kwargs):
        super().__init__()
        assert mode == "L2L" and intraview_negs and isinstance(loss, InfoNCE)
        self.loss = loss

    def forward(self, h1=None, h2=None, **kwargs):
        return dual_branch_l2l_intraview(h1, h2, self.loss.tau)


class SingleBranchContrast(torch.nn.Module):
    def __init__(self, loss, mode, intraview_negs=False, **kwargs):
        super().__init__()
        assert mode == "G2L" and isinstance(loss, JSD)

    def forward(self, h, g, batch=None, hn=None, **kwargs):
        return jsd_g2l(h, g, hn)
