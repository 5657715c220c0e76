"""Link-prediction ranking evaluation (filtered MRR / Hits@k) for the DistMult decoder on the device.

The reference computes no ranking metric; BASELINE.json's north_star asks for MRR / Hits@k,
so this is a build-defined evaluation (the next-row item of SURVEY.md 8f rank 2).  Protocol: for every test
triple (h, r, t) score all N candidate tails (and, for ``side="both"``, all candidate heads), optionally
remove the other KNOWN true candidates of the same query (filtered setting), and rank the true one with the
"realistic" tie rule ``rank = 1 + #greater + #equal / 2``.  The [queries, N] score matrix is never
materialised: ``csrc/rank.hip`` counts inside the MFMA tile epilogue.
"""
from __future__ import annotations

import torch

from . import _lib as L


def _counts(q, z, target, filt=None):
    n_q, h = q.shape
    n = z.size(0)
    dev = q.device
    true = torch.empty(n_q, dtype=torch.float32, device=dev)
    L.call("bkg_rank_true_score_f32", L.ptr(q), q.stride(0), L.ptr(z), z.stride(0), L.ptr(target), n_q, h, L.ptr(true),
           L.stream())
    n_i = (n_q + 63) // 64
    splits = max(1, min(32, 2048 // n_i))
    gt = torch.empty(splits, n_q, dtype=torch.int32, device=dev)
    eq = torch.empty(splits, n_q, dtype=torch.int32, device=dev)
    L.call("bkg_rank_count_f32", L.ptr(q), q.stride(0), n_q, L.ptr(z), z.stride(0), n, h, L.ptr(true), L.ptr(target),
           splits, L.ptr(gt), L.ptr(eq), L.stream())
    gt, eq = gt.sum(0, dtype=torch.int32), eq.sum(0, dtype=torch.int32)
    if filt is not None:
        qptr, cand = filt
        L.call("bkg_rank_filter_f32", L.ptr(q), q.stride(0), L.ptr(z), z.stride(0), h, L.ptr(true), L.ptr(target),
               L.ptr(qptr), L.ptr(cand), n_q, L.ptr(gt), L.ptr(eq), L.stream())
    return gt, eq


def _filter_lists(anchor, rel, known_anchor, known_rel, known_target, num_relations):
    """For each query (anchor, rel): the known true targets, as (qptr [Q+1], cand [total]) int64."""
    key_known = known_anchor * num_relations + known_rel
    order = torch.argsort(key_known)
    key_sorted, tgt_sorted = key_known[order], known_target[order]
    key_q = anchor * num_relations + rel
    lo = torch.searchsorted(key_sorted, key_q, right=False)
    hi = torch.searchsorted(key_sorted, key_q, right=True)
    cnt = hi - lo
    qptr = torch.cat([cnt.new_zeros(1), torch.cumsum(cnt, 0)])
    total = int(qptr[-1].item())
    row = torch.repeat_interleave(torch.arange(anchor.numel(), device=anchor.device), cnt)
    pos = torch.arange(total, device=anchor.device) - qptr[row] + lo[row]
    return qptr.contiguous(), tgt_sorted[pos].contiguous()


@torch.no_grad()
def distmult_ranking(z, rel_emb, edge_index, edge_type, known=None, ks=(1, 3, 10), side="both"):
    """-> ``{"MRR", "Hits@k"..., "mean_rank", "num_queries"}``.

    ``known``: optional ``(edge_index, edge_type)`` of ALL true triples (train + valid + test) for the filtered
    setting.  DistMult is symmetric in head and tail, so head prediction uses ``z[t] * rel[r]`` as the query.
    """
    L.require_cuda(z, rel_emb, edge_index, edge_type)
    z = z.float().contiguous()
    r = rel_emb.size(0)
    ranks = []
    sides = ("tail", "head") if side == "both" else (side,)
    for s in sides:
        anchor, target = (edge_index[0], edge_index[1]) if s == "tail" else (edge_index[1], edge_index[0])
        q = (z[anchor] * rel_emb[edge_type]).contiguous()
        filt = None
        if known is not None:
            k_ei, k_et = known
            k_anchor, k_target = (k_ei[0], k_ei[1]) if s == "tail" else (k_ei[1], k_ei[0])
            filt = _filter_lists(anchor, edge_type, k_anchor, k_et, k_target, r)
        gt, eq = _counts(q, z, target.contiguous(), filt)
        ranks.append(1.0 + gt.double() + eq.double() / 2.0)
    rank = torch.cat(ranks)
    out = {"MRR": float((1.0 / rank).mean()), "mean_rank": float(rank.mean()), "num_queries": int(rank.numel())}
    for k in ks:
        out[f"Hits@{k}"] = float((rank <= k).double().mean())
    return out


def _l1_counts(q, zn, target, filt=None):
    n_q, d = q.shape
    n, dev = zn.size(0), q.device
    true = torch.empty(n_q, dtype=torch.float32, device=dev)
    L.call("bkg_rank_l1_true_f32", L.ptr(q), q.stride(0), L.ptr(zn), zn.stride(0), L.ptr(target), n_q, d, L.ptr(true),
           L.stream())
    n_qb = (n_q + 7) // 8
    splits = max(1, min(64, 4096 // n_qb))
    gt = torch.empty(splits, n_q, dtype=torch.int32, device=dev)
    eq = torch.empty(splits, n_q, dtype=torch.int32, device=dev)
    L.call("bkg_rank_l1_count_f32", L.ptr(q), q.stride(0), n_q, L.ptr(zn), zn.stride(0), n, d, L.ptr(true),
           L.ptr(target), splits, L.ptr(gt), L.ptr(eq), L.stream())
    gt, eq = gt.sum(0, dtype=torch.int32), eq.sum(0, dtype=torch.int32)
    if filt is not None:
        qptr, cand = filt
        L.call("bkg_rank_l1_filter_f32", L.ptr(q), q.stride(0), L.ptr(zn), zn.stride(0), d, L.ptr(true), L.ptr(target),
               L.ptr(qptr), L.ptr(cand), n_q, L.ptr(gt), L.ptr(eq), L.stream())
    return gt, eq


@torch.no_grad()
def transe_ranking(z, rel_emb, edge_index, edge_type, known=None, ks=(1, 3, 10), side="both"):
    """Filtered MRR / Hits@k for the TransE decoder (reference decoder.py:31-38: rows L1-normalised with eps
    1e-12, score = -||h^ + r - t^||_1).  Tail prediction ranks by ``-||(h^ + r) - z^_j||_1``, head prediction by
    ``-||(t^ - r) - z^_j||_1`` (csrc/rank_l1.hip).  Same protocol and tie rule as :func:`distmult_ranking`."""
    L.require_cuda(z, rel_emb, edge_index, edge_type)
    z = z.float()
    zn = (z / z.abs().sum(dim=1, keepdim=True).clamp_min(1e-12)).contiguous()
    r = rel_emb.size(0)
    ranks = []
    for s in (("tail", "head") if side == "both" else (side,)):
        if s == "tail":
            anchor, target = edge_index[0], edge_index[1]
            q = (zn[anchor] + rel_emb[edge_type]).contiguous()
        else:
            anchor, target = edge_index[1], edge_index[0]
            q = (zn[anchor] - rel_emb[edge_type]).contiguous()
        filt = None
        if known is not None:
            k_ei, k_et = known
            k_anchor, k_target = (k_ei[0], k_ei[1]) if s == "tail" else (k_ei[1], k_ei[0])
            filt = _filter_lists(anchor, edge_type, k_anchor, k_et, k_target, r)
        gt, eq = _l1_counts(q, zn, target.contiguous(), filt)
        ranks.append(1.0 + gt.double() + eq.double() / 2.0)
    rank = torch.cat(ranks)
    out = {"MRR": float((1.0 / rank).mean()), "mean_rank": float(rank.mean()), "num_queries": int(rank.numel())}
    for k in ks:
        out[f"Hits@{k}"] = float((rank <= k).double().mean())
    return out
