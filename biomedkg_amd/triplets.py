"""Graph construction from a triplet table (columns ``x_type, x_name, relation, y_type, y_name``) with the
reference's conventions (``biomedkg/data/dataset/_base.py:23-89`` + PyG ``HeteroData.to_homogeneous``,
SURVEY.md Appendix A.10; 8f rank 4), so that node ids / relation ids of a graph built here match the
reference's and its checkpoints and exported embeddings line up:

* node types in ALPHABETICAL order of the raw type names (``np.unique``); inside a type, names sorted;
  homogeneous node id = offset of the type + rank of the name;
* relation ids in order of FIRST APPEARANCE in the table;
* only ONE (head type, tail type) direction per relation is kept -- the first distinct (x_type, relation, y_type)
  row (``triples[0]``); edges keep table order inside a relation and relations are concatenated in
  id order;
* type keys are reduced to letters (``clean_name``): types / (head, relation, tail) keys that collide after
  cleaning overwrite each other, exactly as assigning into the reference's ``HeteroData`` does.

Host-side (pandas); the result feeds the device path as ``edge_index int64 [2,E]`` / ``edge_type int64 [E]``.
"""
from __future__ import annotations

import re

import numpy as np
import torch


def clean_name(s: str) -> str:
    """reference biomedkg/common.py:8-11"""
    return "".join(re.findall("[a-zA-Z]+", s))


def build_graph(df, encoder):
    """-> dict(x, edge_index, edge_type, node_type, num_edge_types, edge_map_index, node_list).
    ``encoder(list_of_names) -> tensor [n, ...]`` supplies the node features (reference ``self.encoder``)."""
    node_types = np.unique(np.concatenate([df["x_type"].unique(), df["y_type"].unique()]))
    relations = df["relation"].unique()
    node_to_index, node_list = {}, []
    node_stores = {}                       # cleaned type name -> x   (insertion ordered; collisions overwrite)
    for nt in node_types:
        names = sorted(set(df.loc[df["x_type"] == nt, "x_name"].values) | set(df.loc[df["y_type"] == nt, "y_name"].values))
        node_list.extend(names)
        node_to_index[nt] = {name: i for i, name in enumerate(names)}
        node_stores[clean_name(nt)] = encoder(names)
    edge_stores, edge_map_index = {}, {}
    for rid, rel in enumerate(relations):
        rdf = df[df["relation"] == rel]
        head, _, tail = rdf[["x_type", "relation", "y_type"]].drop_duplicates().values[0]
        pairs = rdf[(rdf["x_type"] == head) & (rdf["y_type"] == tail)]
        src = [node_to_index[head][n] for n in pairs["x_name"]]
        dst = [node_to_index[tail][n] for n in pairs["y_name"]]
        edge_stores[(clean_name(head), clean_name(rel), clean_name(tail))] = torch.tensor([src, dst], dtype=torch.long).reshape(2, -1)
        edge_map_index[rid] = rel
    offsets, xs, node_type, base = {}, [], [], 0
    for i, (name, x) in enumerate(node_stores.items()):
        offsets[name] = base
        xs.append(x)
        node_type.append(torch.full((x.size(0),), i, dtype=torch.long))
        base += x.size(0)
    eis, ets = [], []
    for i, ((h, _, t), ei) in enumerate(edge_stores.items()):
        eis.append(torch.stack([ei[0] + offsets[h], ei[1] + offsets[t]]))
        ets.append(torch.full((ei.size(1),), i, dtype=torch.long))
    return dict(x=torch.cat(xs, 0), node_type=torch.cat(node_type),
                edge_index=torch.cat(eis, 1) if eis else torch.zeros(2, 0, dtype=torch.long),
                edge_type=torch.cat(ets) if ets else torch.zeros(0, dtype=torch.long),
                num_edge_types=len(edge_stores), edge_map_index=edge_map_index, node_list=node_list)


def to_undirected(edge_index, edge_type, num_nodes=None, reduce="add"):
    """PyG 2.5.3 ``T.ToUndirected()`` as the reference's DPI module applies it (``data_module.py:178``; defaults
    ``reduce="add", merge=True``): append the reversed edges with their types, then ``coalesce`` -- sort by
    ``row * N + col``, merge duplicate pairs.  ``reduce="add"`` is PyG's default and what the reference runs: the edge
    attributes of merged entries are SUMMED, ``edge_type`` included (a pair listed twice under relation r comes out as
    2 r).  In the reference's DPI graph that is harmless -- one relation, id 0 -- and it is kept here for parity;
    ``reduce="first"`` keeps the type of the first listed entry instead (ids stay valid on multi-relation input)."""
    ei = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    et = torch.cat([edge_type, edge_type])
    n = int(num_nodes) if num_nodes is not None else (int(ei.max()) + 1 if ei.numel() else 0)
    key = ei[0] * n + ei[1]
    uniq, inv = torch.unique(key, return_inverse=True)
    if reduce == "add":
        merged = torch.zeros(uniq.numel(), dtype=et.dtype).index_add_(0, inv, et)
    elif reduce == "first":
        first = torch.full((uniq.numel(),), key.numel(), dtype=torch.long).scatter_reduce_(0, inv, torch.arange(key.numel()), "amin")
        merged = et[first]
    else:
        raise ValueError(reduce)
    return torch.stack([uniq // n, uniq % n]), merged
