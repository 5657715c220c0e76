"""Device-side mini-batch samplers with the semantics of the two PyG loaders the reference trains with
(``biomedkg/data_module.py:89-107``): ``NeighborLoader(num_neighbors=[30]*3)`` for GCL and
``GraphSAINTRandomWalkSampler(walk_length=10)`` for KGE.  The draws happen in ``csrc/sample.hip`` over the CSRs
the hot path already keeps in HBM; the bookkeeping around them (unique / relabel) is torch glue on the device.
Every batch satisfies the batch contract of SURVEY.md 8b and never touches the host.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .graph import graph_of


class SubgraphBatch:
    """``.x .edge_index .edge_type .batch_size .n_id .e_id`` (PyG ``Data``-like; already on the device)."""

    def __init__(self, x, edge_index, edge_type, batch_size, n_id, e_id):
        self.x, self.edge_index, self.edge_type = x, edge_index, edge_type
        self.batch_size, self.n_id, self.e_id = batch_size, n_id, e_id

    def to(self, device):
        return self


def rows_of(table, n_id):
    """``table[n_id]``, tagged with where it came from: modules whose first op is a row-wise constant function of the
    features (the never-trained AttentionFusion, ``biomedkg_amd/utils/fusion.py``) apply it to the table once and gather."""
    x = table.index_select(0, n_id)
    x._bkg_rows_of = (table, n_id)
    return x


def _tape_words(tape, shape):
    """A replay tape as the kernels read it: contiguous 32-bit words on the device (None stays None)."""
    if tape is None:
        return None
    L.require_cuda(tape)
    if tape.element_size() != 4 or tuple(tape.shape) != tuple(shape) or tape.is_floating_point():
        raise ValueError(f"replay tape: expected {tuple(shape)} 32-bit integer words, got {tuple(tape.shape)} {tape.dtype}")
    return tape.contiguous()


def _relabel(n_id, num_nodes, *ids):
    lut = torch.full((num_nodes,), -1, dtype=torch.int64, device=n_id.device)
    lut[n_id] = torch.arange(n_id.numel(), device=n_id.device)
    return [lut[i] for i in ids]


class NeighborSampler:
    """Hop-wise uniform sampling of at most ``num_neighbors[h]`` distinct in-edges per frontier node; every node
    is expanded once, when first discovered; node order = seeds, then each hop's new nodes (PyG NeighborLoader,
    ``directed=True, replace=False``)."""

    def __init__(self, edge_index, num_nodes, num_neighbors=(30, 30, 30), seed=0):
        L.require_cuda(edge_index)
        self.edge_index = edge_index          # (the structure cached on the tensor only holds it weakly)
        self.graph = graph_of(edge_index, num_nodes)
        self.num_nodes, self.num_neighbors, self.seed, self.calls = int(num_nodes), tuple(num_neighbors), int(seed), 0

    def _state(self, dev):
        """Fixed-capacity device state of the one-read sampler (csrc/sample.hip: bkg_neighbor_*)."""
        st = getattr(self, "_st", None)
        if st is None:
            n, fan = self.num_nodes, max(min(k if k > 0 else 64, 64) for k in self.num_neighbors)
            i32 = lambda k: torch.empty(k, dtype=torch.int32, device=dev)      # noqa: E731
            i64 = lambda k: torch.empty(k, dtype=torch.int64, device=dev)      # noqa: E731
            cap = n * fan
            flags, offs = i32(n + cap), i32(n + cap)     # the slots' flags right behind the nodes': one scan per hop takes both
            st = self._st = dict(fan=fan, cap=cap, seen=i32(n), n_id=i64(n), ctr=i64(3), pos=i32(cap), new_flag=flags[:n],
                                 node_offs=offs[:n], eflag=flags[n:], edge_offs=offs[n:], lut=i32(n), e_src=i64(cap),
                                 e_dst=i64(cap), e_id=i64(cap),
                                 ws=torch.empty(L.lib().bkg_neighbor_workspace_bytes(n, fan), dtype=torch.uint8, device=dev))
        return st

    def one_read_ok(self):
        return self.num_nodes * max(min(k if k > 0 else 64, 64) for k in self.num_neighbors) < (1 << 31)

    def sample(self, seeds, tape=None):
        """One host read per batch (node and edge counts) instead of three per hop; the same batch, draw for draw, as
        ``sample_hop_by_hop``.  ``tape`` (parity tests): ``[num_nodes, 64]`` raw 32-bit words (int32 / uint32 storage)
        that replace the Philox stream -- node v's Floyd step s reads ``tape[v, s]`` (oracle/samplers.py reads the same)."""
        tape = _tape_words(tape, (self.num_nodes, 64))
        if not self.one_read_ok() or seeds.numel() == 0:
            return self.sample_hop_by_hop(seeds, tape)
        st = self.enqueue(seeds, tape)
        _, nn, ne = st["ctr"].tolist()                                       # the one host read of the batch
        ei = torch.stack([st["e_src"][:ne], st["e_dst"][:ne]])
        ei._bkg_dst_sorted = ei._version      # frontier nodes are expanded in local-id order, hop after hop: targets ascend (graph.Graph)
        return st["n_id"][:nn].clone(), ei, st["e_id"][:ne].clone()

    def enqueue(self, seeds, tape=None):
        """The batch of ``sample`` left in the fixed-capacity device state (``ctr[1]`` nodes in ``n_id``, ``ctr[2]`` edges
        in ``e_src / e_dst / e_id``, already relabelled) -- no host read; biomedkg_amd/sampled_gcl.py trains on it as is."""
        st, csr = self._state(seeds.device), self.graph.by_dst
        self.calls += 1
        seeds = seeds.contiguous()
        L.call("bkg_neighbor_begin", L.ptr(seeds), seeds.numel(), self.num_nodes, L.ptr(st["seen"]), L.ptr(st["n_id"]),
               L.ptr(st["ctr"]), L.stream())
        for hop, k in enumerate(self.num_neighbors):
            fan = min(k if k > 0 else 64, 64)
            L.call("bkg_neighbor_hop", L.ptr(csr.rowptr), L.ptr(csr.col), L.ptr(csr.perm), self.num_nodes, fan,
                   self.seed * 1_000_003 + self.calls, hop, L.ptr(tape), L.ptr(st["seen"]), L.ptr(st["n_id"]), L.ptr(st["ctr"]),
                   L.ptr(st["pos"]), L.ptr(st["new_flag"]), L.ptr(st["node_offs"]), L.ptr(st["eflag"]),
                   L.ptr(st["edge_offs"]), st["cap"], L.ptr(st["e_src"]), L.ptr(st["e_dst"]), L.ptr(st["e_id"]),
                   L.ptr(st["ws"]), st["ws"].numel(), L.stream())
        L.call("bkg_neighbor_finish", self.num_nodes, L.ptr(st["n_id"]), L.ptr(st["ctr"]), L.ptr(st["lut"]), st["cap"],
               L.ptr(st["e_src"]), L.ptr(st["e_dst"]), L.stream())
        return st

    def sample_hop_by_hop(self, seeds, tape=None):
        g, dev = self.graph, seeds.device
        csr = g.by_dst
        self.calls += 1
        seen = torch.zeros(self.num_nodes, dtype=torch.bool, device=dev)
        seen[seeds] = True
        nodes, frontier = [seeds], seeds
        srcs, dsts, eids = [], [], []
        for hop, k in enumerate(self.num_neighbors):
            if frontier.numel() == 0:
                break
            fan = min(k if k > 0 else 64, 64)
            pos = torch.empty(frontier.numel(), fan, dtype=torch.int32, device=dev)
            L.call("bkg_sample_neighbors", L.ptr(csr.rowptr), L.ptr(frontier), frontier.numel(), fan,
                   self.seed * 1_000_003 + self.calls, hop, L.ptr(tape), L.ptr(pos), L.stream())
            valid = pos >= 0
            p = pos[valid].long()
            src = csr.col[p].long()
            dst = frontier.unsqueeze(1).expand_as(pos)[valid]
            srcs.append(src); dsts.append(dst); eids.append(csr.perm[p].long())
            new = torch.unique(src[~seen[src]])
            seen[new] = True
            nodes.append(new)
            frontier = new
        n_id = torch.cat(nodes)
        src, dst = (torch.cat(srcs), torch.cat(dsts)) if srcs else (seeds.new_empty(0), seeds.new_empty(0))
        e_id = torch.cat(eids) if eids else seeds.new_empty(0)
        s, d = _relabel(n_id, self.num_nodes, src, dst)
        return n_id, torch.stack([s, d]), e_id


class SaintRandomWalkSampler:
    """``batch_size`` uniform root nodes, one ``walk_length``-step uniform random walk each over the out-edges,
    node set = the distinct visited nodes (ascending), subgraph = ALL edges among them (GraphSAINT "RW" sampler as
    the reference configures it: ``sample_coverage=0``, so no normalisation statistics)."""

    def __init__(self, edge_index, num_nodes, batch_size, walk_length=10, seed=0):
        L.require_cuda(edge_index)
        self.edge_index = edge_index
        self.graph = graph_of(edge_index, num_nodes)
        self.num_nodes, self.batch_size, self.walk_length = int(num_nodes), int(batch_size), int(walk_length)
        self.seed, self.calls = int(seed), 0

    def walks(self, start, tape=None):
        csr = self.graph.by_src
        self.calls += 1
        tape = _tape_words(tape, (start.numel(), self.walk_length))
        out = torch.empty(start.numel(), self.walk_length + 1, dtype=torch.int64, device=start.device)
        L.call("bkg_random_walk", L.ptr(csr.rowptr), L.ptr(csr.col), L.ptr(start), start.numel(), self.walk_length,
               self.seed * 1_000_003 + self.calls, L.ptr(tape), L.ptr(out), L.stream())
        return out

    _EDGE_CAP = 1 << 22      # induced edges the one-call path holds (more: the torch-glue path below)

    def _state(self, dev):
        st = getattr(self, "_st", None)
        if st is None:
            n, e = self.num_nodes, self.edge_index.size(1)
            cap, ncap = min(e, self._EDGE_CAP), self.batch_size * (self.walk_length + 1) + 1
            i32 = lambda k: torch.empty(k, dtype=torch.int32, device=dev)      # noqa: E731
            i64 = lambda k: torch.empty(k, dtype=torch.int64, device=dev)      # noqa: E731
            st = self._st = dict(cap=cap, ncap=ncap, walks=i64(self.batch_size * (self.walk_length + 1)), sizes=i64(3),
                                 node=[i32(n) for _ in range(3)], edge=[i32(e) for _ in range(2)], n_id=i64(ncap),
                                 out=[i64(cap) for _ in range(4)],
                                 ws=torch.empty(L.lib().bkg_saint_batch_workspace_bytes(n, e), dtype=torch.uint8, device=dev))
        return st

    def sample(self, start=None, tape=None):
        """Walks, node set, induced edges and relabelling in ONE library call (csrc/sample.hip: bkg_saint_batch) and
        one host read of the two sizes; the same batch as ``sample_with_torch_glue``.  ``tape`` (parity tests; needs
        ``start``): ``[walks, walk_length]`` raw 32-bit words replacing the Philox stream (oracle/samplers.py)."""
        dev = self.edge_index.device
        if tape is not None and start is None:
            raise ValueError("a replayed tape needs explicit roots")
        if self.edge_index.size(1) == 0 or self.num_nodes >= (1 << 31):
            return self.sample_with_torch_glue(start, tape)
        if tape is not None:
            tape = _tape_words(tape, (start.numel(), self.walk_length))
        if start is None:
            start = torch.randint(0, self.num_nodes, (self.batch_size,), device=dev)
        st, csr = self._state(dev), self.graph.by_src
        self.calls += 1
        start = start.contiguous()
        L.call("bkg_saint_batch", L.ptr(csr.rowptr), L.ptr(csr.col), L.ptr(start), start.numel(), self.walk_length,
               self.seed * 1_000_003 + self.calls, None, L.ptr(tape), L.ptr(st["walks"]), L.ptr(self.edge_index[0]),
               L.ptr(self.edge_index[1]), None, self.num_nodes, self.edge_index.size(1), L.ptr(st["node"][0]),
               L.ptr(st["node"][1]), L.ptr(st["node"][2]), L.ptr(st["edge"][0]), L.ptr(st["edge"][1]), st["ncap"], st["cap"],
               st["ncap"] - 1, 1, L.ptr(st["n_id"]), L.ptr(st["out"][0]), L.ptr(st["out"][1]), L.ptr(st["out"][2]),
               L.ptr(st["out"][3]), L.ptr(st["sizes"]), L.ptr(st["ws"]), st["ws"].numel(), L.stream())
        nn, ne, _ = st["sizes"].tolist()                                     # the one host read of the batch
        if ne > st["cap"]:
            self.calls -= 1
            return self.sample_with_torch_glue(start, tape)
        return (st["n_id"][:nn].clone(), torch.stack([st["out"][0][:ne], st["out"][1][:ne]]), st["out"][3][:ne].clone())

    def sample_with_torch_glue(self, start=None, tape=None):
        dev = self.edge_index.device
        if start is None:
            start = torch.randint(0, self.num_nodes, (self.batch_size,), device=dev)
        n_id = torch.unique(self.walks(start, tape))
        mask = torch.zeros(self.num_nodes, dtype=torch.bool, device=dev)
        mask[n_id] = True
        src, dst = self.edge_index[0], self.edge_index[1]
        e_id = torch.nonzero(mask[src] & mask[dst]).squeeze(1)
        s, d = _relabel(n_id, self.num_nodes, src[e_id], dst[e_id])
        return n_id, torch.stack([s, d]), e_id


class NeighborBatchLoader:
    """Iterates seed batches (optionally shuffled) -> ``SubgraphBatch`` (reference ``train_dataloader("neighbor")``)."""

    def __init__(self, x, edge_index, edge_type, batch_size, num_neighbors=(30, 30, 30), shuffle=True, seed=0):
        self.x, self.edge_type, self.batch_size, self.shuffle = x, edge_type, int(batch_size), shuffle
        self.sampler = NeighborSampler(edge_index, x.size(0), num_neighbors, seed)

        self.rank, self.world, self.epoch, self.seed = 0, 1, 0, int(seed)

    def set_rank(self, rank, world):
        """DistributedSampler semantics: one shuffled order shared by the ranks (same generator seed), rank r takes
        seed batches r, r + world, ...; every rank runs the same number of steps (the tail wraps around)."""
        self.rank, self.world = int(rank), int(world)

    def _num_batches(self):
        return (self.x.size(0) + self.batch_size - 1) // self.batch_size

    def __len__(self):
        return (self._num_batches() + self.world - 1) // self.world

    def __iter__(self):
        n, dev = self.x.size(0), self.x.device
        if self.world > 1:
            g = torch.Generator().manual_seed(self.seed + self.epoch)      # identical on every rank
            order = (torch.randperm(n, generator=g) if self.shuffle else torch.arange(n)).to(dev)
            self.epoch += 1
        else:
            order = torch.randperm(n, device=dev) if self.shuffle else torch.arange(n, device=dev)
        nb = self._num_batches()
        for step in range(len(self)):
            i = ((step * self.world + self.rank) % nb) * self.batch_size
            seeds = order[i:i + self.batch_size]
            n_id, ei, e_id = self.sampler.sample(seeds)
            et = self.edge_type[e_id] if self.edge_type is not None else None
            yield SubgraphBatch(rows_of(self.x, n_id), ei, et, seeds.numel(), n_id, e_id)


class SaintBatchLoader:
    """``num_steps`` GraphSAINT random-walk subgraphs per epoch (reference ``train_dataloader("saint")``)."""

    def __init__(self, x, edge_index, edge_type, batch_size, walk_length=10, num_steps=1000, seed=0):
        self.x, self.edge_type, self.num_steps = x, edge_type, int(num_steps)
        self.sampler = SaintRandomWalkSampler(edge_index, x.size(0), batch_size, walk_length, seed)

    def set_rank(self, rank, world):
        """Each rank walks from its own roots with its own stream (the steps per epoch stay ``num_steps`` per rank, as
        under Lightning DDP, where GraphSAINT loaders are not sharded but each process draws independently)."""
        self.sampler.seed = self.sampler.seed * 131 + int(rank) + 1

    def __len__(self):
        return self.num_steps

    def __iter__(self):
        for _ in range(self.num_steps):
            n_id, ei, e_id = self.sampler.sample()
            et = self.edge_type[e_id] if self.edge_type is not None else None
            yield SubgraphBatch(rows_of(self.x, n_id), ei, et, n_id.numel(), n_id, e_id)
