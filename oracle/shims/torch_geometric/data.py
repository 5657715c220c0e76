"""Shim of ``torch_geometric.data.HeteroData`` -- only what the reference's ``TripletBase`` touches
(``biomedkg/data/dataset/_base.py:31,58,87-89``): per-node-type ``.x``, per-edge-type ``.edge_index`` and
``to_homogeneous()`` with PyG 2.5.3 semantics: node types concatenated in insertion
order, ``edge_type`` = index of the edge type in insertion order, plus a ``node_type`` vector."""
import types

import torch


class _Store(types.SimpleNamespace):
    pass


class HeteroData:
    def __init__(self):
        self._nodes, self._edges = {}, {}

    def __getitem__(self, key):
        store = self._edges if isinstance(key, tuple) else self._nodes
        return store.setdefault(key, _Store())

    def to_homogeneous(self):
        offset, xs, node_type, base = {}, [], [], 0
        for i, (name, st) in enumerate(self._nodes.items()):
            offset[name] = base
            xs.append(st.x)
            node_type.append(torch.full((st.x.size(0),), i, dtype=torch.long))
            base += st.x.size(0)
        eis, ets = [], []
        for i, ((h, _, t), st) in enumerate(self._edges.items()):
            ei = st.edge_index.clone()
            ei[0] += offset[h]
            ei[1] += offset[t]
            eis.append(ei)
            ets.append(torch.full((ei.size(1),), i, dtype=torch.long))
        out = _Store(x=torch.cat(xs, 0), node_type=torch.cat(node_type),
                     edge_index=torch.cat(eis, 1) if eis else torch.zeros(2, 0, dtype=torch.long),
                     edge_type=torch.cat(ets) if ets else torch.zeros(0, dtype=torch.long))
        out.num_edge_types = len(self._edges)
        out.num_nodes = base
        return out
