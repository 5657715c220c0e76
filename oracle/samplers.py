"""CPU restatement of the two PyG mini-batch loaders the reference trains with -- TEST INFRASTRUCTURE (see
oracle/__init__.py): only tests/, smoke() and bench.py's cpu_baseline leg may import this.

* ``neighbor_sample``  -- ``NeighborLoader(data, num_neighbors=[30] * 3, batch_size=.., shuffle=..)`` as
  ``/root/reference/biomedkg/data_module.py:89-99`` builds it (defaults ``replace=False``, ``directed=True``,
  ``disjoint=False``), i.e. pyg-lib's ``neighbor_sample`` over the CSC of the graph;
* ``saint_random_walk_batch`` -- ``GraphSAINTRandomWalkSampler(data, batch_size, walk_length=10, num_steps=..)`` of
  ``/root/reference/biomedkg/data_module.py:100-107`` (``sample_coverage=0``: no normalisation statistics), i.e.
  torch-sparse's ``random_walk`` + ``saint_subgraph``.

**Parity unpinned by the reference**: it has no tests or fixtures for its loaders, and torch_geometric
2.5.3 / pyg-lib / torch-sparse are not installed here and cannot be (no network) -- this file restates their PUBLISHED
algorithms:

pyg-lib ``neighbor_sample`` (homogeneous, ``replace=False``, ``directed=True``, non-disjoint):
  nodes = seeds; ``begin, end = 0, len(seeds)``; for every hop's fan-out k: for each node v in ``nodes[begin:end]`` IN
  ORDER: its in-neighbours are the CSC column of v (``to_csc``: a stable sort of the edges by target, so a column lists
  the in-edges in edge order); if ``deg <= k`` (or k < 0) every in-edge is taken, in order; else k DISTINCT positions are
  drawn by Robert Floyd's algorithm -- ``for j in range(deg - k, deg): t = uniform integer in [0, j]; take t unless it was
  taken already, then take j`` -- and taken in draw order; each taken edge (u -> v) is recorded with its original edge id,
  and u joins ``nodes`` the first time it is seen (a global-to-local map: every node appears ONCE and is expanded ONCE,
  in the hop after its discovery); then ``begin, end = end, len(nodes)``.  Output: ``nodes`` (seeds first, then in
  order of discovery), the edges in local ids, the edge ids.

torch-sparse ``random_walk(rowptr, col, start, walk_length)``: per walk, per step: ``deg == 0`` -> stay; else step to
  ``col[rowptr[cur] + uniform integer in [0, deg)]`` over the CSR by SOURCE (rows list the out-edges in edge order).
``GraphSAINTRandomWalkSampler``: roots ``randint(0, N, (batch_size,))``; ``node_idx = walks.flatten().unique()``
  (ascending); the batch is ``saint_subgraph(node_idx)``: EVERY edge whose two endpoints are in the set, relabelled.

Randomness.  Upstream draws from host generators whose streams no device can reproduce; here every uniform integer
below n is ``(word * n) >> 32`` of a raw 32-bit word taken from an explicit TAPE, and ``csrc/sample.hip`` maps the same
words the same way when it is handed the tape (its replay mode): walk w, step s reads ``tape[w, s]``; node v's Floyd step
s reads ``tape[v, s]`` -- keyed by the node, which is expanded once, so the tape means the same whatever order a sampler
keeps its frontier in (upstream: discovery order; the device sampler: ascending ids per hop -- a relabelling of the same
batch, which is what the parity tests compare modulo).
"""
from __future__ import annotations

import torch

TAPE_STRIDE = 64          # words per node of a neighbour tape (csrc/sample.hip TAPE_STRIDE: the largest fan-out)


def _below(word: int, n: int) -> int:
    """Uniform integer in [0, n) from a raw 32-bit word: multiply-shift (csrc/sample.hip ``below``)."""
    return ((int(word) & 0xFFFFFFFF) * n) >> 32


def make_tape(rows: int, cols: int, seed: int) -> torch.Tensor:
    """``[rows, cols]`` raw 32-bit words (stored as int32: torch's uint32 support is partial) from a seeded generator."""
    g = torch.Generator().manual_seed(seed)
    return torch.randint(-(1 << 31), 1 << 31, (rows, cols), generator=g, dtype=torch.int64).to(torch.int32)


def csc_of(edge_index, num_nodes):
    """PyG ``to_csc``: (colptr, row, perm) -- the in-edges of every node in (stable) edge order."""
    dst = edge_index[1]
    perm = torch.argsort(dst, stable=True)
    colptr = torch.zeros(num_nodes + 1, dtype=torch.int64)
    colptr[1:] = torch.cumsum(torch.bincount(dst, minlength=num_nodes), 0)
    return colptr, edge_index[0][perm], perm


def csr_of(edge_index, num_nodes):
    """(rowptr, col, perm) by SOURCE, out-edges in (stable) edge order (torch-sparse SparseTensor row-major storage for
    a graph whose edges are already grouped arbitrarily: the order inside a row is the stable one)."""
    src = edge_index[0]
    perm = torch.argsort(src, stable=True)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(src, minlength=num_nodes), 0)
    return rowptr, edge_index[1][perm], perm


def floyd_positions(deg: int, k: int, words) -> list:
    """k distinct positions of [0, deg) in draw order (pyg-lib's without-replacement branch); ``words``: the node's tape."""
    picked = []
    for step, j in enumerate(range(deg - k, deg)):
        t = _below(words[step], j + 1)
        picked.append(j if t in picked else t)
    return picked


def neighbor_sample(edge_index, num_nodes, seeds, num_neighbors, tape):
    """-> dict(n_id [nodes], edge_index [2, edges] local ids (source, target), e_id [edges], hops: list of the node ids
    discovered per hop (hops[0] = seeds)).  ``tape``: ``[num_nodes, 64]`` words (make_tape)."""
    colptr, row, perm = (t.tolist() for t in csc_of(edge_index, num_nodes))
    tape_l = tape.tolist()
    nodes = [int(s) for s in seeds.tolist()]
    local = {v: i for i, v in enumerate(nodes)}
    if len(local) != len(nodes):
        raise ValueError("duplicate seeds")
    src, dst, eid = [], [], []
    hops = [list(nodes)]
    begin, end = 0, len(nodes)
    for k in num_neighbors:
        for i in range(begin, end):
            v = nodes[i]
            beg, deg = colptr[v], colptr[v + 1] - colptr[v]
            if k < 0 or deg <= k:
                pos = list(range(deg))
            else:
                pos = floyd_positions(deg, k, tape_l[v])
            for p in pos:
                u = row[beg + p]
                if u not in local:
                    local[u] = len(nodes)
                    nodes.append(u)
                src.append(local[u]); dst.append(i); eid.append(perm[beg + p])
        begin, end = end, len(nodes)
        hops.append(nodes[begin:end])
    return dict(n_id=torch.tensor(nodes, dtype=torch.int64),
                edge_index=torch.tensor([src, dst], dtype=torch.int64).reshape(2, -1),
                e_id=torch.tensor(eid, dtype=torch.int64), hops=hops)


def random_walk(edge_index, num_nodes, start, walk_length, tape):
    """``[walks, walk_length + 1]`` node ids; ``tape``: ``[walks, walk_length]`` words."""
    rowptr, col, _ = (t.tolist() for t in csr_of(edge_index, num_nodes))
    tape_l = tape.tolist()
    out = []
    for w, cur in enumerate(start.tolist()):
        walk = [cur]
        for s in range(walk_length):
            beg, deg = rowptr[cur], rowptr[cur + 1] - rowptr[cur]
            if deg > 0:
                cur = col[beg + _below(tape_l[w][s], deg)]
            walk.append(cur)
        out.append(walk)
    return torch.tensor(out, dtype=torch.int64).reshape(len(out), walk_length + 1)


def saint_random_walk_batch(edge_index, num_nodes, start, walk_length, tape):
    """-> dict(walks, n_id (ascending), edge_index [2, edges] local ids, e_id (ascending: edge order)) -- every edge among
    the walked nodes."""
    walks = random_walk(edge_index, num_nodes, start, walk_length, tape)
    n_id = torch.unique(walks)
    inside = torch.zeros(num_nodes, dtype=torch.bool)
    inside[n_id] = True
    e_id = torch.nonzero(inside[edge_index[0]] & inside[edge_index[1]]).squeeze(1)
    lut = torch.full((num_nodes,), -1, dtype=torch.int64)
    lut[n_id] = torch.arange(n_id.numel())
    return dict(walks=walks, n_id=n_id, edge_index=lut[edge_index[:, e_id]], e_id=e_id)
