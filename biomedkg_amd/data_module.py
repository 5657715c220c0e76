"""The reference's data modules (``biomedkg/data_module.py``) over this build's pieces, plus a synthetic stand-in.

``PrimeKGModule`` / ``DPIModule`` -- the reference's constructor keywords, attributes and loader methods
(``data_module.py:27-145`` / ``:148-259``): a triplet table on disk (``{data_dir}/kg.csv`` / the DPI csv) ->
``triplets.build_graph`` (the reference's ``TripletBase`` + ``to_homogeneous`` conventions) -> (``to_undirected`` for DPI)
-> ``link_split`` (PyG ``RandomLinkSplit(neg_sampling_ratio=0)``) -> the device mini-batch loaders of ``samplers.py``
(``NeighborLoader([30] * 3)`` / ``GraphSAINTRandomWalkSampler(walk_length=10)`` semantics on the GPU).  Node features:
``random`` (xavier-normal per node type, the reference's ``RandomEncode``), ``lm`` / ``gcl`` (pickled
``{name: array}`` dictionaries in the reference's on-disk layout, ``data/node.py``).  The random draws happen in the
reference's order -- one ``xavier_normal_`` per node type in ``np.unique`` order, then one ``randperm(E)`` -- so that under
the same seed the graph, features and splits are those of the reference (tests/test_data_modules.py against a fixture
recorded from the reference's own classes).

``SyntheticKGModule`` -- the same interface without a download: every "batch" is the whole (split) graph unless
``sampler="device"``; batch contract of SURVEY.md 8b (``.x``, ``.edge_index`` int64 [2,E], ``.edge_type`` int64 [E],
``.batch_size``, ``.to(device)``).
"""
from __future__ import annotations

import glob
import os
import pickle
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


# ---------------------------------------------------------------------------------------- node features
class RandomEncode:
    """reference data/node.py:148-162: ``xavier_normal_`` over the ``[n, embed_dim]`` block of ONE node type."""

    def __init__(self, embed_dim: int = 768):
        self.embed_dim = embed_dim
        self.random_init_ratio = 1

    def __call__(self, lst_node):
        return torch.nn.init.xavier_normal_(torch.empty(len(lst_node), self.embed_dim))


class _PickledEncode:
    """``{node name: array [rows, embed_dim]}`` from a pickle in the reference's layout; a name that is missing gets an
    ``xavier_normal_`` ``[rows, embed_dim]`` block (data/node.py:31-45, 178-191)."""

    rows = 1

    def __init__(self, artifact_path, embed_dim):
        self.artifact_path, self.embed_dim = artifact_path, embed_dim
        self.node_mapping = self.load()
        self.random_init_ratio = 0

    def load(self):
        if not os.path.exists(self.artifact_path):
            self._get_embeddings()
        with open(self.artifact_path, "rb") as file:
            return pickle.load(file)

    def _get_embeddings(self):
        raise FileNotFoundError(self.artifact_path)

    def __call__(self, lst_node):
        out, random_init = [], 0
        for name in lst_node:
            emb = self.node_mapping.get(name, None)
            if emb is None:
                emb = torch.nn.init.xavier_normal_(torch.empty(self.rows, self.embed_dim))
                random_init += 1
            out.append(torch.as_tensor(emb))
        self.random_init_ratio = random_init / max(len(lst_node), 1)
        return torch.stack(out, dim=0)


class LMMultiModalsEncode(_PickledEncode):
    """reference data/node.py:20-58: the per-modality language-model embeddings ``[2, embed_dim]`` per node, read from
    ``data/embed/{modality config stem}_lm.pickle``.  Producing that file runs four HuggingFace encoders over the
    modality tables (node.py:60-145): stage A of the reference pipeline, outside the training hot path (out
    of scope) -- so a missing file is an error that says where the reference writes it."""

    rows = 2

    def __init__(self, config_file: str, embed_dim: int = 768, batch_size: int = 128):
        stem = os.path.splitext(os.path.basename(config_file))[0]
        self.config_file, self.batch_size = config_file, batch_size
        super().__init__(os.path.join("data", "embed", f"{stem}_lm.pickle"), embed_dim)

    def _get_embeddings(self):
        raise FileNotFoundError(
            f"{self.artifact_path}: the language-model node features are not there.  The reference writes this pickle "
            f"({{name: array [2, {self.embed_dim}]}}) with its LMMultiModalsEncode (biomedkg/data/node.py:60-145, HuggingFace "
            "encoders over the modality tables); run that stage with the reference, or use node_init_method=random")

    def __call__(self, lst_node):
        return super().__call__(lst_node).to(torch.float)


class GCLEncode(_PickledEncode):
    """reference data/node.py:165-241: the exported GCL embeddings ``[1, embed_dim]`` per node from
    ``data/gcl_embed/{model}_{fuse}.pickle``; when the file is missing it is produced from the GCL checkpoints under
    ``ckpt/gcl/{gene,drug,disease}/{model}*{fuse}*lm*/*.ckpt`` -- ``export.node_embeddings``: the reference's per-node
    ``NeighborLoader([-1], batch_size=1)`` loop as a few large launches."""

    data_gcl = os.path.join("data", "gcl_embed")
    gcl_ckpt = os.path.join("ckpt", "gcl")

    def __init__(self, model_name: str, fuse_method: str, embed_dim: int):
        self.model_name, self.fuse_method = model_name, fuse_method
        super().__init__(os.path.join(self.data_gcl, f"{model_name}_{fuse_method}.pickle"), embed_dim)

    def _get_embeddings(self):
        from . import export, gcl_module
        classes = {"dgi": gcl_module.DGIModule, "grace": gcl_module.GRACEModule, "ggd": gcl_module.GGDModule}
        if self.model_name not in classes:
            raise NotImplementedError(self.model_name)
        mapping = {}
        for node_type in ["gene", "drug", "disease"]:
            pattern = f"{self.gcl_ckpt}/{node_type}/{self.model_name}*{self.fuse_method}*lm*/*.ckpt"
            files = sorted(glob.glob(pattern))
            if not files:
                raise FileNotFoundError(f"{self.artifact_path} is missing and no checkpoint matches {pattern}")
            module = classes[self.model_name].load_from_checkpoint(files[0])
            data = PrimeKGModule(data_dir="./data/primekg", embed_dim=768, batch_size=128, val_ratio=0.2, test_ratio=0.2,
                                 node_type=["gene/protein" if node_type == "gene" else node_type], node_init_method="lm")
            data.setup(stage="split")
            dev = "cuda"
            module = module.to(dev)
            z = export.node_embeddings(module, data.data.x.to(dev), data.data.edge_index.to(dev))
            mapping.update(export.to_mapping(data.primekg.node_list, z))
        os.makedirs(self.data_gcl, exist_ok=True)
        with open(self.artifact_path, "wb") as file:
            pickle.dump(mapping, file, protocol=pickle.HIGHEST_PROTOCOL)


def get_node_encode_method(node_init_method, embed_dim, model_name, fuse_method, modality_config_path):
    """reference data_module.py:8-24 (an unknown method yields None there too)."""
    if node_init_method is None or node_init_method == "random":
        return RandomEncode(embed_dim=embed_dim)
    if node_init_method == "lm":
        return LMMultiModalsEncode(config_file=modality_config_path, embed_dim=embed_dim)
    if node_init_method == "gcl":
        return GCLEncode(model_name=model_name, fuse_method=fuse_method, embed_dim=embed_dim)
    return None


# ---------------------------------------------------------------------------------------- graphs from triplet tables
class GraphData:
    """What ``HeteroData.to_homogeneous()`` hands the reference's modules, as far as they read it: ``x, edge_index,
    edge_type, node_type, num_nodes, num_edge_types`` (+ ``edge_label / edge_label_index`` on a split)."""

    def __init__(self, x, edge_index, edge_type, node_type=None, num_edge_types=None, **more):
        self.x, self.edge_index, self.edge_type, self.node_type = x, edge_index, edge_type, node_type
        self.num_nodes = int(x.size(0))
        self.num_edge_types = int(num_edge_types if num_edge_types is not None
                                  else (int(edge_type.max()) + 1 if edge_type.numel() else 0))
        self.batch_size = self.num_nodes
        for k, v in more.items():
            setattr(self, k, v)

    def to(self, device):
        out = GraphData.__new__(GraphData)
        for k, v in vars(self).items():
            setattr(out, k, v.to(device) if torch.is_tensor(v) else v)
        return out


class TripletDataset:
    """reference data/dataset/_base.py:13-21: ``.df``, ``.data``, ``.edge_map_index``, ``.node_list``."""

    def __init__(self, df, encoder):
        from .triplets import build_graph
        self.df, self.encoder = df, encoder
        g = build_graph(df, encoder)
        self.data = GraphData(g["x"], g["edge_index"], g["edge_type"], g["node_type"], g["num_edge_types"])
        self.edge_map_index, self.node_list = g["edge_map_index"], g["node_list"]


class PrimeKG(TripletDataset):
    """reference data/dataset/primekg.py:9-43: ``{data_dir}/kg.csv`` filtered to the rows whose two node types are both
    in ``node_type``.  The reference downloads the file when it is missing (``wget`` from the Harvard Dataverse, or the
    ``tdc`` package); there is no network on the boxes this build targets, so a missing file is an error that says so."""

    URL = "https://dataverse.harvard.edu/api/access/datafile/6180620"

    def __init__(self, data_dir, node_type, encoder=None):
        import pandas as pd
        csv_path = f"{data_dir}/kg.csv"
        if not os.path.exists(csv_path):
            raise FileNotFoundError(f"{csv_path}: PrimeKG's kg.csv is not there (the reference fetches it from {self.URL}; "
                                    "put the file in place, or run with data=synthetic)")
        df = pd.read_csv(csv_path, low_memory=False)
        if node_type:
            df = df[df["x_type"].isin(list(node_type)) & df["y_type"].isin(list(node_type))]
        super().__init__(df=df, encoder=encoder)


class DPI(TripletDataset):
    """reference data/dataset/dpi.py:8-19: the csv AT ``data_dir`` (a file path), rows with a missing value dropped."""

    def __init__(self, data_dir, encoder=None):
        import pandas as pd
        if not os.path.exists(data_dir):
            raise FileNotFoundError(f"{data_dir}: the DPI benchmark csv is not there (columns x_type, x_name, relation, "
                                    "y_type, y_name)")
        super().__init__(df=pd.read_csv(data_dir).dropna(), encoder=encoder)


def link_split(data, val_ratio, test_ratio):
    """PyG 2.5.3 ``RandomLinkSplit(num_val, num_test, neg_sampling_ratio=0.0)`` as the reference calls it
    (data_module.py:65-69; ``is_undirected=False``): ONE ``torch.randperm(E)`` from the global generator, float ratios
    -> ``int(ratio * E)``; the train graph holds the first ``E - n_val - n_test`` permuted edges, the val graph the SAME
    edges, the test graph train + val edges; ``edge_type`` travels with its edges; ``edge_label`` (ones) /
    ``edge_label_index`` = each split's supervision edges (the modules never read them)."""
    e = data.edge_index.size(1)
    perm = torch.randperm(e)
    n_val = int(val_ratio * e) if isinstance(val_ratio, float) else int(val_ratio)
    n_test = int(test_ratio * e) if isinstance(test_ratio, float) else int(test_ratio)
    n_train = e - n_val - n_test
    if n_train <= 0:
        raise ValueError("Insufficient number of edges for training")
    parts = (perm[:n_train], perm[n_train:n_train + n_val], perm[n_train + n_val:])
    graphs = (perm[:n_train], perm[:n_train], perm[:n_train + n_val])
    out = []
    for msg, sup in zip(graphs, parts):
        out.append(GraphData(data.x, data.edge_index[:, msg], data.edge_type[msg], data.node_type, data.num_edge_types,
                             edge_label=torch.ones(sup.numel()), edge_label_index=data.edge_index[:, sup]))
    return tuple(out)


class _StarLoader:
    """``NeighborLoader(num_neighbors=[-1], batch_size=1, shuffle=False)`` (reference ``subgraph_dataloader``): batch v =
    node v first, then its distinct in-neighbours in edge order, and every edge INTO v.  (The export path does not loop
    over this -- ``export.node_embeddings`` runs the same stars in bulk.)"""

    def __init__(self, data):
        from .samplers import SubgraphBatch
        self.data, self.Batch = data, SubgraphBatch
        dst = data.edge_index[1]
        self.order = torch.argsort(dst, stable=True)
        self.ptr = torch.zeros(data.num_nodes + 1, dtype=torch.long, device=dst.device)
        self.ptr[1:] = torch.cumsum(torch.bincount(dst, minlength=data.num_nodes), 0)

    def __len__(self):
        return self.data.num_nodes

    def __iter__(self):
        d = self.data
        ptr = self.ptr.tolist()
        for v in range(d.num_nodes):
            e_id = self.order[ptr[v]:ptr[v + 1]]
            src = d.edge_index[0, e_id]
            seen, n_id = {v: 0}, [v]
            for s in src.tolist():
                if s not in seen:
                    seen[s] = len(n_id)
                    n_id.append(s)
            n_id = torch.tensor(n_id, dtype=torch.long, device=src.device)
            ei = torch.stack([torch.tensor([seen[s] for s in src.tolist()], dtype=torch.long, device=src.device),
                              torch.zeros(e_id.numel(), dtype=torch.long, device=src.device)])
            yield self.Batch(d.x[n_id], ei, d.edge_type[e_id], 1, n_id, e_id)


class _TripletModule:
    """What ``PrimeKGModule`` and ``DPIModule`` share: the split and the loaders (reference data_module.py:58-145 and its
    copy :175-259).  The loaders sample ON THE DEVICE (``samplers.py``); there is no host fallback."""

    device = "cuda"

    def _split(self, stage):
        if stage == "split":
            self.train_data, self.val_data, self.test_data = link_split(self.data, self.val_ratio, self.test_ratio)

    def _on_device(self, data):
        hit = getattr(data, "_bkg_on_device", None)
        if hit is None:
            hit = data._bkg_on_device = data.to(self.device)
        return hit

    def _seed(self):
        return int(torch.initial_seed() & 0x7FFFFFFF)

    def _neighbor(self, data, shuffle):
        from .samplers import NeighborBatchLoader
        d = self._on_device(data)
        return NeighborBatchLoader(d.x, d.edge_index, d.edge_type, self.batch_size, (30, 30, 30), shuffle, self._seed())

    def _saint(self, data, num_steps):
        from .samplers import SaintBatchLoader
        d = self._on_device(data)
        return SaintBatchLoader(d.x, d.edge_index, d.edge_type, self.batch_size, walk_length=10, num_steps=num_steps,
                                seed=self._seed())

    def subgraph_dataloader(self):
        return _StarLoader(self.data)

    def all_dataloader(self):
        return self._neighbor(self.data, False)

    def train_dataloader(self, loader_type: str = "neighbor"):
        assert loader_type in ["neighbor", "saint"]
        return self._neighbor(self.train_data, True) if loader_type == "neighbor" else self._saint(self.train_data, 1000)

    def val_dataloader(self, loader_type: str = "neighbor"):
        assert loader_type in ["neighbor", "saint"]
        return self._neighbor(self.val_data, False) if loader_type == "neighbor" else self._saint(self.val_data, 100)

    def test_dataloader(self, loader_type: str = "neighbor"):
        assert loader_type in ["neighbor", "saint"]
        return self._neighbor(self.test_data, False) if loader_type == "neighbor" else self._saint(self.test_data, 100)


class PrimeKGModule(_TripletModule):
    """reference data_module.py:27-145."""

    def __init__(self, data_dir: str, embed_dim: int, node_type, batch_size: int, val_ratio: float, test_ratio: float,
                 node_init_method: str = None, gcl_model: str = None, gcl_fuse_method: str = None):
        self.hparams = types.SimpleNamespace(data_dir=data_dir, embed_dim=embed_dim, node_type=node_type, batch_size=batch_size,
                                             val_ratio=val_ratio, test_ratio=test_ratio, node_init_method=node_init_method,
                                             gcl_model=gcl_model, gcl_fuse_method=gcl_fuse_method)
        self.data_dir, self.node_type = data_dir, node_type
        self.val_ratio, self.test_ratio, self.batch_size = val_ratio, test_ratio, batch_size
        self.encoder = get_node_encode_method(node_init_method=node_init_method, embed_dim=embed_dim, model_name=gcl_model,
                                              fuse_method=gcl_fuse_method,
                                              modality_config_path="configs/lm_modality/primekg_modality.yaml")

    def setup(self, stage: str = "split"):
        self.primekg = PrimeKG(data_dir=self.data_dir, node_type=self.node_type, encoder=self.encoder)
        self.edge_map_index = self.primekg.edge_map_index
        self.data = self.primekg.data
        self._split(stage)


class DPIModule(_TripletModule):
    """reference data_module.py:148-259: the DPI table, made undirected (PyG ``ToUndirected()``: reversed edges appended,
    duplicate pairs merged) before the split."""

    def __init__(self, data_dir: str, embed_dim: int, batch_size: int, val_ratio: float, test_ratio: float,
                 node_init_method: str = None, gcl_model: str = None, gcl_fuse_method: str = None):
        self.hparams = types.SimpleNamespace(data_dir=data_dir, embed_dim=embed_dim, batch_size=batch_size, val_ratio=val_ratio,
                                             test_ratio=test_ratio, node_init_method=node_init_method, gcl_model=gcl_model,
                                             gcl_fuse_method=gcl_fuse_method)
        self.data_dir = data_dir
        self.val_ratio, self.test_ratio, self.batch_size = val_ratio, test_ratio, batch_size
        self.encoder = get_node_encode_method(node_init_method=node_init_method, embed_dim=embed_dim, model_name=gcl_model,
                                              fuse_method=gcl_fuse_method,
                                              modality_config_path="configs/lm_modality/dpi_modality.yaml")

    def setup(self, stage: str = "split"):
        from .triplets import to_undirected
        self.dpi = DPI(data_dir=self.data_dir, encoder=self.encoder)
        self.edge_map_index = self.dpi.edge_map_index
        d = self.dpi.data
        ei, et = to_undirected(d.edge_index, d.edge_type, num_nodes=d.num_nodes)
        # PyG's ToUndirected SUMS the edge_type of merged duplicate pairs (reduce="add", what the reference runs): on the
        # reference's one-relation DPI table every id stays 0, but a multi-relation table could come out with ids >=
        # num_edge_types, which PyG's RGCNConv / the decoder's rel_emb lookup answer with an IndexError -- and the device
        # kernels, which index rel_emb / rel_ptr unchecked, with an out-of-bounds read.  Same contract, raised here.
        if et.numel() and int(et.max()) >= int(d.num_edge_types):
            raise IndexError(f"DPIModule: ToUndirected(reduce='add') merged duplicate pairs into relation id {int(et.max())} "
                             f">= num_edge_types = {int(d.num_edge_types)} (a pair listed under several relations; "
                             "triplets.to_undirected(..., reduce='first') keeps valid ids)")
        self.data = GraphData(d.x, ei, et, d.node_type, d.num_edge_types)
        self._split(stage)
