"""Synthetic PrimeKG++-shaped graphs for bench.py and the full-size property tests (there is no
network for the real dataset).

G-full mimics what the reference's ``PrimeKGModule`` builds (``biomedkg/data/dataset/_base.py:23-89``
-> ``to_homogeneous()``): node ids grouped by node type (10 types, PrimeKG-like sizes), 30 relation
types each tied to one (source type, target type) pair with a skewed histogram (two relations carry
~37 % and ~33 % of the edges, a long tail below 0.1 %), Zipf-like endpoint popularity inside a type
(weight ~ rank^-0.8) and a hub block for the drug-drug relation (a few thousand nodes with in-degree
in the hundreds to thousands).  G-small is the single-type 5k/50k GCL graph of BASELINE configs 1-2.
"""
from __future__ import annotations

import numpy as np
import torch

# PrimeKG-like node-type sizes (sum = 129,375 at scale 1.0), alphabetical as to_homogeneous() orders them
NODE_TYPES = [
    ("anatomy", 14035), ("biological_process", 28642), ("cellular_component", 4176), ("disease", 17080),
    ("drug", 7957), ("effect/phenotype", 15311), ("exposure", 818), ("gene/protein", 27671),
    ("molecular_function", 11169), ("pathway", 2516),
]

# (name, src type, dst type, share of edges).  Shares are renormalised.
RELATIONS = [
    ("anatomy_protein_present", 0, 7, 0.37), ("drug_drug", 4, 4, 0.33), ("protein_protein", 7, 7, 0.08),
    ("disease_phenotype_positive", 3, 5, 0.037), ("bioprocess_protein", 1, 7, 0.036),
    ("cellcomp_protein", 2, 7, 0.02), ("disease_protein", 3, 7, 0.02), ("molfunc_protein", 8, 7, 0.017),
    ("drug_effect", 4, 5, 0.016), ("bioprocess_bioprocess", 1, 1, 0.013), ("pathway_protein", 9, 7, 0.0105),
    ("disease_disease", 3, 3, 0.008), ("contraindication", 4, 3, 0.0075), ("drug_protein", 4, 7, 0.0063),
    ("anatomy_protein_absent", 0, 7, 0.0049), ("phenotype_phenotype", 5, 5, 0.0046),
    ("anatomy_anatomy", 0, 0, 0.0035), ("molfunc_molfunc", 8, 8, 0.0033), ("indication", 4, 3, 0.0023),
    ("cellcomp_cellcomp", 2, 2, 0.0012), ("phenotype_protein", 5, 7, 0.0008), ("off-label use", 4, 3, 0.0006),
    ("pathway_pathway", 9, 9, 0.0006), ("exposure_disease", 6, 3, 0.0006), ("exposure_exposure", 6, 6, 0.0005),
    ("exposure_bioprocess", 6, 1, 0.0004), ("exposure_protein", 6, 7, 0.0003),
    ("disease_phenotype_negative", 3, 5, 0.0003), ("exposure_molfunc", 6, 8, 0.00002),
    ("exposure_cellcomp", 6, 2, 0.00001),
]


def _zipf_pick(rng, lo, size, count, alpha=0.8):
    w = np.arange(1, size + 1, dtype=np.float64) ** (-alpha)
    w /= w.sum()
    ranks = rng.choice(size, size=count, p=w)
    perm = rng.permutation(size)          # popularity is not tied to the id order
    return lo + perm[ranks]


def primekg_like(num_nodes=130_000, num_edges=4_000_000, num_relations=30, seed=42):
    """-> dict(edge_index int64 [2,E], edge_type int64 [E], num_nodes, num_relations, node_type)."""
    rng = np.random.default_rng(seed)
    total = sum(s for _, s in NODE_TYPES)
    sizes = [max(2, int(round(s * num_nodes / total))) for _, s in NODE_TYPES]
    sizes[int(np.argmax(sizes))] += num_nodes - sum(sizes)   # rounding slack goes to the largest type
    if min(sizes) < 1:
        raise ValueError("num_nodes too small for the 10 PrimeKG node types")
    offs = np.concatenate([[0], np.cumsum(sizes)])
    rels = RELATIONS[:num_relations]
    share = np.array([r[3] for r in rels])
    share /= share.sum()
    counts = np.maximum(1, np.floor(share * num_edges)).astype(np.int64)
    counts[0] += num_edges - counts.sum()
    src, dst, typ = [], [], []
    for rid, ((name, st, dt, _), c) in enumerate(zip(rels, counts)):
        s = _zipf_pick(rng, offs[st], sizes[st], c)
        if name == "drug_drug":  # hub block: most targets fall on ~3k drugs (in-degree 500-3000 at full scale)
            hubs = min(sizes[dt], max(8, int(3000 * num_nodes / 130_000)))
            d = offs[dt] + rng.permutation(sizes[dt])[:hubs][rng.integers(0, hubs, size=c)]
        else:
            d = _zipf_pick(rng, offs[dt], sizes[dt], c)
        src.append(s); dst.append(d); typ.append(np.full(c, rid, dtype=np.int64))
    src, dst, typ = np.concatenate(src), np.concatenate(dst), np.concatenate(typ)
    order = rng.permutation(src.size)     # edges arrive in no particular order
    edge_index = torch.from_numpy(np.stack([src[order], dst[order]]).astype(np.int64))
    node_type = torch.from_numpy(np.repeat(np.arange(len(sizes)), sizes).astype(np.int64))
    return dict(edge_index=edge_index, edge_type=torch.from_numpy(typ[order]), num_nodes=int(num_nodes),
                num_relations=len(rels), node_type=node_type)


def gcl_small(num_nodes=5_000, num_edges=50_000, seed=42):
    """Single-type GCL graph (BASELINE configs 1-2): Chung-Lu, weights ~ rank^-0.8, no self loops,
    de-duplicated; ``edge_index int64 [2, <=E]``."""
    rng = np.random.default_rng(seed)
    num_edges = min(int(num_edges), num_nodes * (num_nodes - 1))      # no more distinct ordered pairs than the graph has
    w = np.arange(1, num_nodes + 1, dtype=np.float64) ** (-0.8)
    w /= w.sum()
    got = np.empty((0, 2), dtype=np.int64)
    for _ in range(200):                                              # (a nearly complete tiny graph: settle for what came)
        if got.shape[0] >= num_edges:
            break
        need = int((num_edges - got.shape[0]) * 1.3) + 16
        cand = np.stack([rng.choice(num_nodes, need, p=w), rng.choice(num_nodes, need, p=w)], 1)
        cand = cand[cand[:, 0] != cand[:, 1]]
        got = np.unique(np.concatenate([got, cand]), axis=0)
    got = got[rng.permutation(got.shape[0])[:num_edges]]
    return torch.from_numpy(got.T.copy())


def lm_features(num_nodes, modalities=2, dim=768, seed=42):
    """``[n, M, D]`` features L2-normalised ACROSS the modality axis per (node, feature), as the
    reference's LMMultiModalsEncode does (``biomedkg/data/node.py:116-117``)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(num_nodes, modalities, dim, generator=g)
    return x / x.norm(dim=1, keepdim=True)
