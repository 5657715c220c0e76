"""Memory-lean form of the CPU oracle's two training steps -- TEST INFRASTRUCTURE (see oracle/__init__.py): what
bench.py's ``cpu_baseline`` leg times at the FULL bench size (SURVEY.md 8d "CPU baseline": "a checkpointed oracle variant
proven equal on G-small").

The plain oracle (``ref_modules`` over ``pyg_ops``) keeps, like PyG's CPU path, every per-relation ``[N, d]`` aggregate,
every ``[E, d]`` gathered message table and every ``[T, d]`` decoder temporary alive for the backward: at N = 130,000 /
E = 4,000,000 / R = 30 its training step needs more than 62 GB (OOM-killed).  Here the SAME arithmetic, in
the same order, runs with those temporaries recomputed in the backward instead of stored
(``torch.utils.checkpoint``):

* RGCNConv (``pyg_ops.rgcn_conv``): one checkpoint per relation term ``mean_r(x) @ W_r`` -- saved: ``x`` and the edge
  selection, nothing of size ``[N, d]`` per relation;
* GCNConv (``pyg_ops.gcn_conv``): the propagate ``index_add_(x W^T [src] * w)`` is one checkpoint -- saved: ``x W^T``;
* decoder + BCE (``ref_modules.kge_step``): triplets in chunks of ``CHUNK``, each chunk's ``sum`` of loss terms one
  checkpoint -- saved: ``z`` and the ids.

Checkpointing changes no value: forward results are bit-identical to the plain oracle's except where a chunked sum
replaces one long sum (the BCE mean: chunk partial sums added up, <= 1e-6 relative), and the gradients are those of the
same graph of operations.  ``tests/test_oracle_lean.py`` holds it to ``ref_modules`` on the 5,000-node graph (losses 1e-6,
every gradient 1e-5 in the Frobenius norm) with the same recorded draws.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from . import pyg_ops as pyg
from . import rand as _rand

CHUNK = 1 << 20          # triplets per decoder chunk


# --------------------------------------------------------------------------- conv layers
def _rel_term(x, w_r, src, dst):
    """One relation of ``pyg_ops.rgcn_conv``: scatter-mean of the sources' rows, then ``@ W_r``."""
    n = x.size(0)
    s = torch.zeros(n, x.size(1), dtype=x.dtype).index_add_(0, dst, x.index_select(0, src))
    cnt = torch.zeros(n, dtype=x.dtype).index_add_(0, dst, torch.ones(dst.numel(), dtype=x.dtype))
    return (s / cnt.clamp(min=1).unsqueeze(1)) @ w_r


def rgcn_conv(x, edge_index, edge_type, weight, root, bias):
    """``pyg_ops.rgcn_conv``, relation terms checkpointed."""
    out = torch.zeros(x.size(0), weight.size(2), dtype=x.dtype)
    for r in range(weight.size(0)):
        sel = edge_type == r
        src, dst = edge_index[0, sel], edge_index[1, sel]
        out = out + checkpoint(_rel_term, x, weight[r], src, dst, use_reentrant=False)
    out = out + x @ root
    if bias is not None:
        out = out + bias
    return out


def _propagate(xw, src, dst, w):
    out = torch.zeros_like(xw)
    out.index_add_(0, dst, xw.index_select(0, src) * w.unsqueeze(1))
    return out


def gcn_conv(x, edge_index, weight, bias=None):
    """``pyg_ops.gcn_conv``, the propagate checkpointed."""
    ei, w = pyg.gcn_norm(edge_index, x.size(0), x.dtype)
    out = checkpoint(_propagate, x @ weight.t(), ei[0], ei[1], w, use_reentrant=False)
    if bias is not None:
        out = out + bias
    return out


def _run(stack, conv, x, *graph):
    """``ref_modules._Stack._run`` over the lean conv functions (same ReLU / dropout placement, same draws)."""
    last = len(stack.graph_layers) - 1
    for i, layer in enumerate(stack.graph_layers):
        x = conv(layer, x, *graph)
        if i != last:
            x = torch.relu(x)
            if stack.drop_out:
                x = pyg.dropout(x, 0.2, stack.training)
    return x


def gcn_encoder(enc, x, edge_index):
    return _run(enc, lambda L, h, ei: gcn_conv(h, ei, L.lin.weight, L.bias), x, edge_index)


def rgcn_encoder(enc, x, edge_index, edge_type):
    return _run(enc, lambda L, h, ei, et: rgcn_conv(h, ei, et, L.weight, L.root, L.bias), x, edge_index, edge_type)


# --------------------------------------------------------------------------- the two steps
def ggd_loss(model, x, edge_index):
    """``ref_modules.ggd_loss`` / ``GGD.forward`` (reference gcl.py:73-93, gcl_module.py:229-234)."""
    if _rand.current().coin().item() < model.p:
        x, _ = pyg.mask_feature(x, p=0.4)
        edge_index, _ = pyg.dropout_edge(edge_index, p=0.4)
    pos = gcn_encoder(model.encoder, x, edge_index)
    neg = gcn_encoder(model.encoder, x[_rand.current().perm(x.size(0))], edge_index)
    for layer in model.mlp[:-1]:
        pos, neg = torch.relu(layer(pos)), torch.relu(layer(neg))
    pred = torch.cat([model.mlp[-1](pos).sum(1), model.mlp[-1](neg).sum(1)])
    gt = torch.cat([torch.ones(pos.size(0)), torch.zeros(neg.size(0))])
    return F.binary_cross_entropy_with_logits(pred, gt)


def _chunk_bce_sum(decoder, z, edge_index, edge_type, label):
    s = decoder(z, edge_index, edge_type)
    return F.binary_cross_entropy_with_logits(s, torch.full_like(s, label), reduction="sum")


def kge_step(model, x, edge_index, edge_type, neg_ratio):
    """``ref_modules.kge_step`` (reference kge_module.py:115-129): same negatives and ids (same draws), the decoder +
    BCE over chunks of ``CHUNK`` triplets.  -> (loss, z)."""
    from .ref_modules import sample_neg_edges
    z = rgcn_encoder(model.encoder, x, edge_index, edge_type)
    neg_ei, neg_type = sample_neg_edges(edge_index, edge_type, neg_ratio)
    total = torch.zeros((), dtype=z.dtype)
    count = 0
    for ei, et, label in ((edge_index, edge_type, 1.0), (neg_ei, neg_type, 0.0)):
        for lo in range(0, ei.size(1), CHUNK):
            total = total + checkpoint(_chunk_bce_sum, model.decoder, z, ei[:, lo:lo + CHUNK], et[lo:lo + CHUNK], label,
                                       use_reentrant=False)
        count += ei.size(1)
    reg = z.pow(2).mean() + model.decoder.rel_emb.pow(2).mean()
    return total / count + 1e-2 * reg, z
