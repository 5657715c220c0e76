"""Shim of the two ``torch_geometric.transforms`` the reference's data modules apply
(``biomedkg/data_module.py:65-69,178-185``), restated from PyG 2.5.3 (absent here: parity unpinned, oracle/__init__.py)
for homogeneous data objects with attributes ``x, edge_index, edge_type, node_type`` (oracle/shims/.../data.py).

``RandomLinkSplit(num_val, num_test, neg_sampling_ratio=0.0)`` (torch_geometric/transforms/random_link_split.py,
``is_undirected=False, add_negative_train_samples=True, disjoint_train_ratio=0``): ONE ``torch.randperm(E)`` from the
global generator; float ratios -> ``int(ratio * E)``; train graph = the first ``E - n_val - n_test`` permuted edges, val
graph = the SAME edges, test graph = train + val edges; every edge-level attribute (``edge_type``) is indexed alongside;
each split gets ``edge_label`` (ones: no negatives are drawn at ratio 0) and ``edge_label_index`` (train / val / test
supervision edges).

``ToUndirected()`` (transforms/to_undirected.py, ``reduce="add"``): ``utils.to_undirected`` -- append the reversed edges,
then ``coalesce``: sort by ``row * N + col``, merge duplicate pairs, edge attributes of merged entries are SUMMED (so
``edge_type`` of a pair listed twice is the sum of its ids -- harmless in the reference's DPI graph, whose single relation
has id 0)."""
import copy

import torch


def _is_edge_attr(data, key, value):
    return key != "edge_index" and torch.is_tensor(value) and key.startswith("edge") and value.size(0) == data.edge_index.size(1)


class RandomLinkSplit:
    def __init__(self, num_val=0.1, num_test=0.2, is_undirected=False, key="edge_label", split_labels=False,
                 add_negative_train_samples=True, neg_sampling_ratio=1.0, disjoint_train_ratio=0.0, edge_types=None,
                 rev_edge_types=None):
        assert not is_undirected and neg_sampling_ratio == 0.0 and disjoint_train_ratio == 0.0 and edge_types is None
        self.num_val, self.num_test, self.key = num_val, num_test, key

    def _split(self, src, out, index):
        for k, v in list(vars(src).items()):
            if k == "edge_index":
                setattr(out, k, v[:, index])
            elif _is_edge_attr(src, k, v):
                setattr(out, k, v[index])

    def __call__(self, data):
        train, val, test = copy.copy(data), copy.copy(data), copy.copy(data)
        e = data.edge_index.size(1)
        perm = torch.randperm(e, device=data.edge_index.device)
        num_val = int(self.num_val * e) if isinstance(self.num_val, float) else self.num_val
        num_test = int(self.num_test * e) if isinstance(self.num_test, float) else self.num_test
        num_train = e - num_val - num_test
        if num_train <= 0:
            raise ValueError("Insufficient number of edges for training")
        train_edges, val_edges, test_edges = perm[:num_train], perm[num_train:num_train + num_val], perm[num_train + num_val:]
        self._split(data, train, train_edges)
        self._split(data, val, train_edges)
        self._split(data, test, perm[:num_train + num_val])
        for out, idx in ((train, train_edges), (val, val_edges), (test, test_edges)):
            setattr(out, self.key, torch.ones(idx.numel()))
            setattr(out, self.key + "_index", data.edge_index[:, idx])
        return train, val, test


def to_undirected(edge_index, edge_attrs, num_nodes):
    row, col = torch.cat([edge_index[0], edge_index[1]]), torch.cat([edge_index[1], edge_index[0]])
    attrs = [torch.cat([a, a]) for a in edge_attrs]
    key = row * num_nodes + col
    uniq, inv = torch.unique(key, return_inverse=True)             # sorted by (row, col)
    out = []
    for a in attrs:
        out.append(torch.zeros((uniq.numel(),) + a.shape[1:], dtype=a.dtype).index_add_(0, inv, a))    # reduce="add"
    return torch.stack([uniq // num_nodes, uniq % num_nodes]), out


class ToUndirected:
    def __init__(self, reduce="add", merge=True):
        assert reduce == "add"

    def __call__(self, data):
        keys = [k for k, v in vars(data).items() if _is_edge_attr(data, k, v)]
        n = getattr(data, "num_nodes", None) or data.x.size(0)
        ei, vals = to_undirected(data.edge_index, [getattr(data, k) for k in keys], n)
        data.edge_index = ei
        for k, v in zip(keys, vals):
            setattr(data, k, v)
        return data
