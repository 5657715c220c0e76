"""Embedding export (stage B -> C): what the reference's ``GCLEncode`` / ``KGEEncode`` compute
(``biomedkg/data/node.py:196-241, 285-330``) -- ``model(batch.x, batch.edge_index[, edge_type])[:batch_size]`` over
``NeighborLoader(num_neighbors=[-1], batch_size=1, shuffle=False)``, i.e. for every node v the encoder applied to
v's ONE-HOP star subgraph (v plus its in-neighbours, only the edges into v), although the encoder has four
layers (a reference quirk) -- as a few large launches instead of N tiny ones.

The stars of a block of centre nodes are laid side by side in one union graph: the centres first, then one leaf
copy per in-edge.  A leaf reached by two parallel edges may be copied twice: leaves have no in-edges inside a
star, so every copy carries the same activations at every layer and the centre's sums / means are unchanged.
A self loop stays an edge from the centre to itself.  The result is stored in
the reference's on-disk format: a pickle of ``{node_name: np.ndarray [1, out_dim]}``.
"""
from __future__ import annotations

import pickle

import torch


def star_union(edge_index, edge_type, lo, hi):
    """Union of the one-hop stars of centres [lo, hi) -> (feature row per union node, edge_index, edge_type)."""
    src, dst = edge_index[0], edge_index[1]
    sel = (dst >= lo) & (dst < hi)
    s, d = src[sel], dst[sel] - lo
    loop = s == dst[sel]                   # a self loop stays an edge centre -> centre (RGCNConv averages it in;
    n_c = hi - lo                          # GCNConv's add_remaining_self_loops drops it again)
    leaf = ~loop
    rows = torch.cat([torch.arange(lo, hi, device=edge_index.device), s[leaf]])
    src_u = d.clone()
    src_u[leaf] = n_c + torch.arange(int(leaf.sum()), device=edge_index.device)
    ei = torch.stack([src_u, d])
    return rows, ei.contiguous(), (edge_type[sel].contiguous() if edge_type is not None else None)


def _blocks(edge_index, num_nodes, max_union_nodes):
    """Centre ranges whose union graphs stay under ``max_union_nodes`` nodes."""
    indeg = torch.bincount(edge_index[1], minlength=num_nodes) + 1
    cum = torch.cumsum(indeg, 0).tolist()
    out, lo, base = [], 0, 0
    for v in range(num_nodes):
        if cum[v] - base > max_union_nodes and v > lo:
            out.append((lo, v))
            lo, base = v, cum[v - 1]
    out.append((lo, num_nodes))
    return out


@torch.no_grad()
def node_embeddings(module, x, edge_index, edge_type=None, max_union_nodes=2_000_000):
    """``[N, out_dim]``: row v = ``module(x_star_v, edge_index_star_v[, edge_type_star_v])[0]`` in eval mode."""
    was_training = module.training
    module.eval()
    n = x.size(0)
    outs = []
    for lo, hi in _blocks(edge_index, n, max_union_nodes):
        rows, ei, et = star_union(edge_index, edge_type, lo, hi)
        xb = x[rows]
        z = module(xb, ei) if et is None else module(xb, ei, et)
        outs.append(z[: hi - lo])
    module.train(was_training)
    return torch.cat(outs, dim=0)


def to_mapping(names, embeddings):
    """The reference's in-memory / on-disk layout: ``{name: np.ndarray [1, out_dim]}`` (node.py:238-241)."""
    emb = embeddings.detach().cpu().numpy()
    return {name: emb[i:i + 1] for i, name in enumerate(names)}


def export_pickle(path, names, embeddings):
    with open(path, "wb") as f:
        pickle.dump(to_mapping(names, embeddings), f, protocol=pickle.HIGHEST_PROTOCOL)
