"""Synthetic stand-in with the interface of the reference's data modules
(``biomedkg/data_module.py``: ``setup``, ``{train,val,test}_dataloader(loader_type)``, ``.data``,
``.edge_map_index``), so the entry points run without the PrimeKG download.  Dataset ingestion and
the PyG mini-batch samplers are out of this build's scope (8f rank 1):
every "batch" here is the whole (split) graph, already in the batch contract of SURVEY.md 8b
(``.x``, ``.edge_index`` int64 [2,E], ``.edge_type`` int64 [E], ``.batch_size``, ``.to(device)``).
"""
from __future__ import annotations

import types

import torch

from . import synthetic


class Batch:
    def __init__(self, x, edge_index, edge_type=None):
        self.x, self.edge_index, self.edge_type = x, edge_index, edge_type
        self.batch_size = x.size(0)

    def to(self, device):
        et = self.edge_type.to(device) if self.edge_type is not None else None
        return Batch(self.x.to(device), self.edge_index.to(device), et)


class _Loader:
    def __init__(self, batch, steps):
        self.batch, self.steps = batch, steps

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            yield self.batch


class SyntheticKGModule:
    def __init__(self, data_dir="", embed_dim=768, node_type=None, batch_size=128, val_ratio=0.2, test_ratio=0.2,
                 node_init_method="random", num_nodes=5000, num_edges=50000, num_relations=8, steps_per_epoch=4,
                 sampler="full", gcl_model=None, gcl_fuse_method=None, seed=42):
        self.embed_dim, self.node_init_method = embed_dim, node_init_method
        self.val_ratio, self.test_ratio = val_ratio, test_ratio
        self.num_nodes, self.num_edges, self.num_relations = num_nodes, num_edges, num_relations
        self.steps, self.seed = steps_per_epoch, seed
        self.batch_size, self.sampler = batch_size, sampler   # "full": whole-graph batches; "device": the reference's
        self.single_type = isinstance(node_type, (list, tuple)) and len(node_type) == 1

    def setup(self, stage: str = "split"):
        n = self.num_nodes
        if self.single_type:          # GCL: one node type, no relation types
            ei = synthetic.gcl_small(n, self.num_edges, seed=self.seed)
            et = torch.zeros(ei.size(1), dtype=torch.int64)
            names = ["intra_type"]
        else:
            g = synthetic.primekg_like(n, self.num_edges, self.num_relations, seed=self.seed)
            ei, et = g["edge_index"], g["edge_type"]
            names = [r[0] for r in synthetic.RELATIONS[: g["num_relations"]]]
        gen = torch.Generator().manual_seed(self.seed)
        if self.node_init_method == "lm":
            x = synthetic.lm_features(n, 2, self.embed_dim, seed=self.seed)
        elif self.node_init_method == "gcl":
            x = torch.randn(n, 1, self.embed_dim, generator=gen)
        else:                         # reference RandomEncode: xavier_normal [n, embed_dim]
            x = torch.randn(n, self.embed_dim, generator=gen) * (2.0 / (n + self.embed_dim)) ** 0.5
        # RandomLinkSplit(neg_sampling_ratio=0) message-passing graphs
        perm = torch.randperm(ei.size(1), generator=gen)
        n_val, n_test = int(self.val_ratio * perm.numel()), int(self.test_ratio * perm.numel())
        tr, va = perm[: perm.numel() - n_val - n_test], perm[perm.numel() - n_val - n_test: perm.numel() - n_test]
        self.train_data = Batch(x, ei[:, tr].contiguous(), et[tr].contiguous())
        self.val_data = self.train_data
        tv = torch.cat([tr, va])
        self.test_data = Batch(x, ei[:, tv].contiguous(), et[tv].contiguous())
        self.data = types.SimpleNamespace(num_edge_types=len(names), num_nodes=n)
        self.edge_map_index = {i: name for i, name in enumerate(names)}

    def _loader(self, data, loader_type, steps, shuffle):
        """reference data_module.py:89-145: NeighborLoader([30]*3) for "neighbor", GraphSAINT-RW(10) for "saint";
        with ``sampler="device"`` both run on the GPU (biomedkg_amd/samplers.py), else the whole graph is the batch."""
        assert loader_type in ["neighbor", "saint"]
        if self.sampler != "device":
            return _Loader(data, steps)
        from .samplers import NeighborBatchLoader, SaintBatchLoader
        d = data.to("cuda")
        if loader_type == "neighbor":
            return NeighborBatchLoader(d.x, d.edge_index, d.edge_type, self.batch_size, (30, 30, 30), shuffle, self.seed)
        return SaintBatchLoader(d.x, d.edge_index, d.edge_type, self.batch_size, walk_length=10, num_steps=steps,
                                seed=self.seed)

    def train_dataloader(self, loader_type: str = "neighbor"):
        return self._loader(self.train_data, loader_type, self.steps, True)

    def val_dataloader(self, loader_type: str = "neighbor"):
        return self._loader(self.val_data, loader_type, max(1, self.steps // 10), False)

    def test_dataloader(self, loader_type: str = "neighbor"):
        return self._loader(self.test_data, loader_type, max(1, self.steps // 10), False)
