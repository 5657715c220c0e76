"""Node-partitioned full-graph training across the GPUs of one node (BASELINE configs[3]): rank p owns a
contiguous range of nodes -- chosen so that every rank holds about E/P in-edges -- stores the CSR rows (all in-edges)
and the activations of its nodes, and keeps a full replica of the weights.  Nothing here has a reference counterpart:
the reference's only multi-GPU mode is Lightning DDP over mini-batches (``bench.py --mode replicas``).

Exchange, per conv layer (``Comm``): an ALL-GATHER of the layer's gather table (GCN: after the ``x W^T`` transform, so
the wire carries ``out`` columns; R-GCN: the layer input) and, in the backward pass, the REDUCE-SCATTER of the table's
gradient (fp32 partial tables summed at their owners).  Both are issued asynchronously and waited for where the data is
needed: the R-GCN root product runs under the all-gather, the root / weight-gradient products under the reduce-scatter.
In bf16 mode the tables cross xGMI as bf16 (half the bytes).  Weight gradients are averaged with the flat bucket of
``distributed.GradBucket``; losses are scaled by ``world / global count`` so that this AVERAGE is the gradient of the
whole-graph loss.  NT-Xent is ROW partitioned (``ntxent_partitioned``): the normalised rows and the row sums D are
all-gathered, each rank runs its anchor rows against all columns; because W_ij = E_ij (1/D_i + 1/D_j) is symmetric no
gradient has to be sent back.

Padded slot space: shards are padded to ``n_max`` rows so that the gathered table is one ``[world * n_max, d]`` matrix and
node g lives at ``slot(g) = owner(g) * n_max + (g - lo_owner)``; the local CSRs are built directly over slots, so no
index translation happens per step.  A ``dropout_edge`` mask (GGD / GRACE augmentation) only re-weights the local CSRs
(``PartitionedGraph.gcn_weights(keep)``): degrees are global, every rank derives them from the full edge list it holds
(O(E) integer work, no communication), nothing is re-sorted.

The arithmetic is injected (``backend``): the product passes ``HipBackend`` -- the SAME fused layers as the single-GPU
path (``ops.gcn_conv`` / ``gcn_conv_views`` / ``rgcn_conv``, which take the exchange from the graph object) -- and the
gloo tests pass a torch backend (``tests/cpu_backend.py``) that runs the composable form below on CPU.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


# ----------------------------------------------------------------------------- exchange
class Comm:
    """The collective pair of one partition: padded-shard all-gather and its reduce-scatter, begin / end split so that
    independent kernels run while the bytes cross xGMI (RCCL runs the collective on its own stream; ``end`` makes the
    current stream wait for it)."""

    def __init__(self, world, n_local, n_max, group=None):
        self.world, self.n_local, self.n_max, self.group = int(world), int(n_local), int(n_max), group
        # one rank has nothing to exchange; BKG_PARTITION_FORCE_COLLECTIVES=1 issues the collectives anyway (a 1-rank
        # RCCL group: the way to run the exact collective calls of the N-rank path on a single-GPU box)
        self.live = self.world > 1 or os.environ.get("BKG_PARTITION_FORCE_COLLECTIVES") == "1"

    def shard_buffer(self, d, dtype, device):
        """A padded ``[n_max, d]`` shard for a producer to write its ``n_local`` rows into (the pad rows are never
        referenced by any CSR: their content is irrelevant)."""
        return torch.empty(self.n_max, d, dtype=dtype, device=device)

    def _padded(self, shard):
        if shard.size(0) == self.n_max:
            return shard if shard.is_contiguous() else shard.contiguous()
        buf = self.shard_buffer(shard.size(1), shard.dtype, shard.device)
        buf[: shard.size(0)].copy_(shard)
        return buf

    def all_gather_begin(self, shard):
        shard = self._padded(shard)
        if not self.live:
            return shard, None
        out = torch.empty(self.world * self.n_max, shard.size(1), dtype=shard.dtype, device=shard.device)
        return out, dist.all_gather_into_tensor(out, shard, group=self.group, async_op=True)

    @staticmethod
    def all_gather_end(pending):
        out, work = pending
        if work is not None:
            work.wait()
        return out

    def all_gather(self, shard):
        return self.all_gather_end(self.all_gather_begin(shard))

    def reduce_scatter_begin(self, dtable):
        """``dtable [world * n_max, d]`` (this rank's contribution to every owner) -> the rank's summed ``[n_local, d]``."""
        if not self.live:
            return dtable[: self.n_local], None
        dtable = dtable if dtable.is_contiguous() else dtable.contiguous()
        out = torch.empty(self.n_max, dtable.size(1), dtype=dtable.dtype, device=dtable.device)
        work = dist.reduce_scatter_tensor(out, dtable, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return out[: self.n_local], work

    @staticmethod
    def reduce_scatter_end(pending):
        out, work = pending
        if work is not None:
            work.wait()
        return out

    def reduce_scatter(self, dtable):
        return self.reduce_scatter_end(self.reduce_scatter_begin(dtable))

    def all_gather_vector(self, v_local, fill=0.0):
        """``[n_local]`` (or ``[k, n_local]``) per rank -> ``[k, world * n_max]`` in slot order."""
        v = v_local.reshape(-1, v_local.size(-1))
        buf = torch.full((v.size(0), self.n_max), fill, dtype=v.dtype, device=v.device)
        buf[:, : v.size(1)] = v
        if not self.live:
            return buf
        out = torch.empty(self.world * v.size(0), self.n_max, dtype=v.dtype, device=v.device)
        dist.all_gather_into_tensor(out, buf, group=self.group)
        return out.view(self.world, v.size(0), self.n_max).permute(1, 0, 2).reshape(v.size(0), self.world * self.n_max)


class _AllGatherRows(torch.autograd.Function):
    """forward: all-gather of the rank's padded shard; backward: reduce-scatter (sum) of the table gradient."""

    @staticmethod
    def forward(ctx, shard, comm):
        ctx.comm, ctx.dtype = comm, shard.dtype
        out = comm.all_gather(shard)
        return out.clone() if out.data_ptr() == shard.data_ptr() else out

    @staticmethod
    def backward(ctx, dtable):
        # partial tables are summed in fp32 whatever the table's own precision
        return ctx.comm.reduce_scatter(dtable.float().contiguous()).to(ctx.dtype), None


def all_gather_rows(shard, comm):
    return _AllGatherRows.apply(shard, comm)


# ----------------------------------------------------------------------------- partition
class NodePartition:
    """Contiguous node ranges balanced by ``in-degree + node_weight``: the gather / aggregate kernels cost per in-edge,
    the products, epilogues and row stores per node (about 16 edge-gathers' worth at d = 256), and the floor keeps the
    ranks that own hub nodes from shrinking to a few hundred rows."""

    NODE_WEIGHT = 16.0

    def __init__(self, edge_index, num_nodes, rank, world, node_weight=None):
        self.num_nodes, self.rank, self.world = int(num_nodes), int(rank), int(world)
        dev = edge_index.device
        self.node_weight = float(self.NODE_WEIGHT if node_weight is None else node_weight)
        load = torch.bincount(edge_index[1], minlength=num_nodes).double() + self.node_weight
        cum = torch.cumsum(load, 0)
        targets = cum[-1] * torch.arange(1, world, device=dev, dtype=torch.float64) / world
        cuts = torch.searchsorted(cum, targets).tolist()          # one host sync, at partition time only
        bounds = [0] + [int(c) + 1 for c in cuts] + [self.num_nodes]
        for i in range(1, len(bounds)):                            # keep the ranges non-decreasing and in range
            bounds[i] = min(max(bounds[i], bounds[i - 1]), self.num_nodes)
        self.bounds = bounds
        self.lo, self.hi = bounds[rank], bounds[rank + 1]
        self.n_local = self.hi - self.lo
        self.n_max = max(max(b - a for a, b in zip(bounds[:-1], bounds[1:])), 1)
        self._bounds_t = torch.tensor(bounds, device=dev, dtype=torch.int64)
        self.comm = Comm(world, self.n_local, self.n_max)
        self._valid = None

    def slots(self, ids):
        """global node ids -> rows of the gathered ``[world * n_max, d]`` table"""
        owner = torch.bucketize(ids, self._bounds_t[1:-1], right=True)
        return owner * self.n_max + (ids - self._bounds_t[owner])

    def node_slots(self):
        """slot of every node, in node order ``[N]`` (the gathered table without its padding = ``table[node_slots()]``)"""
        if self._valid is None:
            self._valid = self.slots(torch.arange(self.num_nodes, device=self._bounds_t.device))
        return self._valid

    def is_local(self, ids):
        return (ids >= self.lo) & (ids < self.hi)

    def pad(self, x_local):
        """[n_local, d] -> [n_max, d] (zero rows at the end)"""
        if x_local.size(0) == self.n_max:
            return x_local
        return torch.cat([x_local, x_local.new_zeros(self.n_max - x_local.size(0), x_local.size(1))], dim=0)

    def shard(self, x_full):
        return x_full[self.lo:self.hi]


class PartitionedGraph:
    """The rank's slice of ``A_hat = D^-1/2 (A + I) D^-1/2`` (PyG gcn_norm semantics): rows = its nodes,
    columns = table slots, the unit self loops as explicit entries.  Carries what the fused layers read off a graph
    (``by_dst`` / ``by_src`` / ``n_out`` / ``n_table`` / ``comm``); ``gcn_weights(keep)`` re-weights the same CSRs for a
    ``dropout_edge`` view -- global degrees from the full edge list, no communication, nothing re-sorted."""

    def __init__(self, part: NodePartition, edge_index, backend, keep=None):
        self.part, self.backend, n = part, backend, part.num_nodes
        self.edge_index = edge_index
        src, dst = edge_index[0], edge_index[1]
        mine = part.is_local(dst)
        eid = torch.nonzero(mine).squeeze(1)
        s, d = src[eid], dst[eid]
        own = torch.arange(part.lo, part.hi, device=edge_index.device)
        col = torch.cat([part.slots(s), part.slots(own)])                   # neighbours + the unit self loop
        row = torch.cat([d - part.lo, own - part.lo])
        self._entry = torch.cat([eid, edge_index.size(1) + own])            # entry -> index into cat(w_edge, self_w)
        self.num_nodes = self.n_out = part.n_local
        self.n_table = part.world * part.n_max
        self.comm = part.comm
        self.by_dst = backend.csr(row, col, part.n_local)                    # rows: local targets
        self.by_src = backend.csr(col, row, self.n_table)                    # rows: table slots
        self.fwd, self.bwd = self.by_dst, self.by_src
        self._weights = {}
        self._default_keep = keep
        gw = self.gcn_weights(keep)
        self.w_fwd, self.w_bwd = gw.w_dst, gw.w_src

    def gcn_weights(self, keep=None):
        from .graph import GcnWeights
        if keep is None:
            keep = self._default_keep
        key = None if keep is None else (id(keep), keep._version)
        hit = self._weights.get("kept" if keep is not None else "plain")
        if hit is not None and hit[0] == key:
            return hit[1]
        w_edge, self_w = self.backend.gcn_norm(self.edge_index, self.part.num_nodes, keep)
        w = torch.cat([w_edge, self_w])[self._entry]
        out = GcnWeights(self.backend.permute(self.by_dst, w), self.backend.permute(self.by_src, w), None)
        self._weights["kept" if keep is not None else "plain"] = (key, out, keep)   # (keep kept alive: id() stays unique)
        return out


def gcn_conv_partitioned(x_local, weight, bias, pg: PartitionedGraph, backend, relu=False, drop=None, gw=None):
    """``dropout(relu(A_hat[local rows] @ all_gather(x_local W^T) + b))`` -> ``[n_local, out]`` (composable form)."""
    part = pg.part
    gw = gw if gw is not None else pg.gcn_weights()
    xw = backend.linear(x_local, weight)                                     # transform BEFORE the exchange:
    table = all_gather_rows(xw, part.comm)                                   # the wire carries `out` columns
    return backend.spmm_fn(table, pg.by_dst, pg.by_src, gw.w_dst, gw.w_src, part.n_local, bias=bias, relu=relu, drop=drop)


class PartitionedRelGraph:
    """The rank's slice of the R-GCN mean aggregation (PyG RGCNConv aggr="mean"): (relation, local target)
    segments over table slots, the segment -> target reduction, and their transposes.  With ``HipBackend`` this IS a
    ``graph.RelGraph`` over (slot sources, local targets), i.e. what ``ops.rgcn_conv`` takes."""

    def __init__(self, part: NodePartition, edge_index, edge_type, num_relations, backend):
        self.part, self.num_relations = part, int(num_relations)
        src, dst = edge_index[0], edge_index[1]
        mine = part.is_local(dst)
        s, d, t = part.slots(src[mine]), dst[mine] - part.lo, edge_type[mine]
        self.n_table = part.world * part.n_max
        self.fused = backend.rel_graph(torch.stack([s, d]), t, part.n_local, self.num_relations, self.n_table, part.comm)
        if self.fused is not None:
            return
        key = t * part.n_local + d                                           # segment key, sorted by (rel, target)
        uniq, seg_of_edge, cnt = torch.unique(key, return_inverse=True, return_counts=True)
        self.num_segments = int(uniq.numel())
        seg_dst = uniq % max(part.n_local, 1)
        seg_rel = uniq // max(part.n_local, 1)
        w = 1.0 / cnt[seg_of_edge].to(torch.float32)
        self.seg_fwd = backend.csr(seg_of_edge, s, self.num_segments)        # segment <- table slots (mean)
        self.seg_bwd = backend.csr(s, seg_of_edge, self.n_table)
        self.seg_w_fwd, self.seg_w_bwd = backend.permute(self.seg_fwd, w), backend.permute(self.seg_bwd, w)
        iota = torch.arange(self.num_segments, device=edge_index.device)
        self.dst_fwd = backend.csr(seg_dst, iota, part.n_local)              # local target <- its segments (sum)
        self.dst_bwd = backend.csr(iota, seg_dst, self.num_segments)
        self.rel_ptr = torch.searchsorted(seg_rel, torch.arange(self.num_relations + 1, device=seg_rel.device)
                                          ).to(torch.int32)


def rgcn_conv_partitioned(x_local, weight, root, bias, prg: PartitionedRelGraph, backend, relu=False, drop=None):
    """RGCNConv over the rank's targets (composable form): all-gather the layer input, segment means, per-relation
    products, per-target sum, + root term + bias, ReLU / dropout in the last epilogue."""
    part = prg.part
    table = all_gather_rows(x_local, part.comm)
    h = backend.spmm_fn(table, prg.seg_fwd, prg.seg_bwd, prg.seg_w_fwd, prg.seg_w_bwd, prg.num_segments)
    y = backend.grouped_linear(h, weight, prg.rel_ptr)
    xr = backend.matmul(x_local, root)
    return backend.spmm_fn(y, prg.dst_fwd, prg.dst_bwd, None, None, part.n_local, addend=xr, bias=bias, relu=relu,
                           drop=drop)


# ----------------------------------------------------------------------------- encoders over shards
def _run_stack(layers, x_local, conv, training, drop_out, dropout_of):
    last = len(layers) - 1
    for i, layer in enumerate(layers):
        if i == last:
            x_local = conv(layer, x_local, False, None, False)
        else:
            drop = dropout_of(x_local.size(0), layer.out_channels, x_local.device) if (drop_out and training) else None
            x_local = conv(layer, x_local, True, drop, True)
    return x_local


def gcn_encoder_partitioned(encoder, x_local, pg: PartitionedGraph, backend, dropout_of=None, keep=None):
    """``GCNEncoder.forward`` (reference encoder.py:153-162) on the rank's shard; ``encoder`` is the ordinary
    module (replicated weights ``graph_layers[i].lin.weight / .bias``); ``keep``: a ``dropout_edge`` mask over ALL
    edges (the same on every rank)."""
    gw = pg.gcn_weights(keep)
    if backend.fused:
        from . import ops
        if ops.features_bf16_ok(x_local):
            x_local = ops.cached_bf16(x_local)
        conv = lambda L, x, relu, drop, hidden: ops.gcn_conv(x, L.lin.weight, L.bias, pg, gw, relu=relu, drop=drop,
                                                             hidden=hidden)
    else:
        conv = lambda L, x, relu, drop, hidden: gcn_conv_partitioned(x, L.lin.weight, L.bias, pg, backend, relu, drop, gw)
    return _run_stack(encoder.graph_layers, x_local, conv, encoder.training, encoder.drop_out, dropout_of)


def rgcn_encoder_partitioned(encoder, x_local, prg: PartitionedRelGraph, backend, dropout_of=None):
    """``RGCN.forward`` (reference encoder.py:50-59) on the rank's shard."""
    if prg.fused is not None:
        from . import ops
        if ops.features_bf16_ok(x_local):
            x_local = ops.cached_bf16(x_local)
        conv = lambda L, x, relu, drop, hidden: ops.rgcn_conv(x, L.weight, L.root, L.bias, prg.fused, relu=relu, drop=drop,
                                                              hidden=hidden)
    else:
        conv = lambda L, x, relu, drop, hidden: rgcn_conv_partitioned(x, L.weight, L.root, L.bias, prg, backend, relu, drop)
    return _run_stack(encoder.graph_layers, x_local, conv, encoder.training, encoder.drop_out, dropout_of)


def gather_embeddings(z_local, part: NodePartition):
    """Full ``[world * n_max, d]`` table of the final embeddings (autograd: reduce-scatter of dz) for the KGE
    scorer; index it with ``part.slots(node_ids)``."""
    return all_gather_rows(z_local, part.comm)


# ----------------------------------------------------------------------------- backends
class HipBackend:
    """The product arithmetic: csrc kernels through ``biomedkg_amd.ops`` / ``graph`` -- the fused single-GPU layers,
    which read the exchange off the partitioned graph objects."""

    fused = True

    @staticmethod
    def csr(key, other, n_rows):
        from .graph import CSR
        return CSR(key.to(torch.int64).contiguous(), other.to(torch.int64).contiguous(), n_rows)

    @staticmethod
    def permute(csr, values):
        return csr.permute_f32(values.to(torch.float32).contiguous())

    @staticmethod
    def gcn_norm(edge_index, num_nodes, keep):
        from .graph import graph_of
        return graph_of(edge_index, num_nodes).gcn_norm_raw(keep)

    @staticmethod
    def rel_graph(slot_edges, edge_type, n_local, num_relations, n_table, comm):
        from .graph import RelGraph
        rg = RelGraph(slot_edges.contiguous(), edge_type.contiguous(), n_local, num_relations, n_src=n_table)
        rg.n_out, rg.n_table, rg.comm = n_local, n_table, comm
        return rg

    @staticmethod
    def linear(x, weight, bias=None):
        from . import ops
        return ops.linear(x, weight, bias)

    @staticmethod
    def matmul(x, w):
        from . import ops
        return ops.linear(x, w.t())          # root is stored [in, out]

    @staticmethod
    def spmm_fn(table, fwd, bwd, w_fwd, w_bwd, n_out, addend=None, bias=None, relu=False, drop=None):
        from . import ops
        return ops.spmm_fn(table, fwd, bwd, w_fwd, w_bwd, n_out, addend=addend, bias=bias, relu=relu, drop=drop)

    @staticmethod
    def grouped_linear(h, weight, group_ptr):
        from . import ops
        return ops.grouped_linear(h, weight, group_ptr)


# ----------------------------------------------------------------------------- whole training steps over shards
def _ggd_head(ggd, pos_h, neg_h, backend, total, world):
    for layer in ggd.mlp[:-1]:
        pos_h = torch.relu(backend.linear(pos_h, layer.weight, layer.bias))
        neg_h = torch.relu(backend.linear(neg_h, layer.weight, layer.bias))
    pos_h = backend.linear(pos_h, ggd.mlp[-1].weight, ggd.mlp[-1].bias).sum(1)
    neg_h = backend.linear(neg_h, ggd.mlp[-1].weight, ggd.mlp[-1].bias).sum(1)
    pred = torch.cat([pos_h, neg_h])
    gt = torch.cat([torch.ones_like(pos_h), torch.zeros_like(neg_h)])
    return torch.nn.functional.binary_cross_entropy_with_logits(pred, gt, reduction="sum") * (world / total)


def ggd_loss_partitioned(ggd, x_local, pg: PartitionedGraph, backend, perm, dropout_of=None, keep=None, mask_feature=None):
    """``GGDModule.calculate_loss`` (reference gcl_module.py:229-234, gcl.py:73-93) for the rank's nodes: both encoder
    passes run over the rank's shard, the BCE terms of the local nodes are summed and divided by the GLOBAL count, times
    ``world`` so that the bucket's gradient AVERAGE equals the gradient of the whole-graph loss.  ``x_local``: the rank's
    rows of the (possibly feature-masked) input; ``perm``: the corruption permutation of all N nodes and ``keep``: the
    optional ``dropout_edge`` mask over all E edges (every rank must pass the same ones).  The corrupted pass never
    needs other ranks' input rows: ``x[perm] W^T = (x W^T)[perm]``, so the permutation is applied to the gathered table
    of the FIRST layer's product (the wire carries h columns once for both views)."""
    part = pg.part
    total = 2.0 * part.num_nodes
    enc = ggd.encoder
    gw = pg.gcn_weights(keep)
    perm_slots = torch.arange(pg.n_table, dtype=torch.int64, device=perm.device)   # padding slots map to themselves
    node_slots = part.node_slots()
    perm_slots[node_slots] = node_slots[perm]                               # slot(j) -> slot(perm[j]): a bijection
    layer0 = enc.graph_layers[0]
    if backend.fused:
        from . import ops
        n_loc, n_dec = part.n_local, part.n_max        # the path is chosen from rank-invariant sizes: all ranks alike
        if (ops.bf16_storage() and x_local.dim() == 2 and x_local.size(1) % 256 == 0 and len(enc.graph_layers) >= 2
                and all(ops.views_ok(n_dec, l.in_channels, l.out_channels, 2) for l in enc.graph_layers[1:])
                and ops.views_ok(n_dec, 256, layer0.out_channels, 2)):
            # both views side by side through every kernel (model/gcl.py GGD._forward_views), exchange per layer
            if ops.features_bf16_ok(x_local):
                x_local = ops.cached_bf16(x_local)
            xw = ops.xw_table(x_local, layer0.lin.weight)                   # [n_local, h] bf16
            table = all_gather_rows(xw, part.comm)                          # [n_table, h]: ONE exchange for both views
            h = ops.pair_rows(table, perm_slots)                            # slot j: [xw_j | xw_perm(j)]
            last = len(enc.graph_layers) - 1
            for i, layer in enumerate(enc.graph_layers):
                drop = None
                if i != last and enc.drop_out and enc.training and dropout_of is not None:
                    drop = dropout_of(n_loc, 2 * layer.out_channels, h.device)
                h = ops.gcn_conv_views(h, None if i == 0 else layer.lin.weight, layer.bias, pg, gw, 2, relu=(i != last),
                                       drop=drop, hidden=True, is_table=(i == 0))
            hh = h.view(2 * n_loc, h.size(1) // 2)
            for layer in ggd.mlp[:-1]:
                hh = ops.linear(hh, layer.weight, layer.bias, act=ops.ACT_RELU)
            head = ggd.mlp[-1]
            if ops.rowsum_linear_ok(hh, head.weight):          # (W h + b).sum(1) = h . W.sum(0) + b.sum(), as model/gcl.py
                s = ops.linear_rowsum(hh, head.weight, head.bias).view(n_loc, 2)
            else:
                s = ops.linear(hh, head.weight, head.bias).sum(1).view(n_loc, 2)
            pred = torch.cat([s[:, 0], s[:, 1]])
            gt = torch.cat([torch.ones_like(s[:, 0]), torch.zeros_like(s[:, 1])])
            return torch.nn.functional.binary_cross_entropy_with_logits(pred, gt, reduction="sum") * (part.world / total)
    # composable form: first layer by hand (shared product, permuted table), the rest through the stack
    xw = backend.linear(x_local, layer0.lin.weight)
    table = all_gather_rows(xw, part.comm)
    outs = []
    for t in (table, table[perm_slots]):
        drop = (dropout_of(part.n_local, layer0.out_channels, t.device)
                if (dropout_of is not None and enc.drop_out and enc.training) else None)
        h = backend.spmm_fn(t, pg.by_dst, pg.by_src, gw.w_dst, gw.w_src, part.n_local, bias=layer0.bias, relu=True, drop=drop)
        last = len(enc.graph_layers) - 1
        for i, layer in enumerate(enc.graph_layers[1:], start=1):
            drop = (dropout_of(part.n_local, layer.out_channels, h.device)
                    if (i != last and dropout_of is not None and enc.drop_out and enc.training) else None)
            h = gcn_conv_partitioned(h, layer.lin.weight, layer.bias, pg, backend, relu=(i != last), drop=drop, gw=gw)
        outs.append(h)
    return _ggd_head(ggd, outs[0], outs[1], backend, total, part.world)


def kge_loss_partitioned(model, x_local, prg: PartitionedRelGraph, backend, pos_ei, pos_et, neg_ei, neg_et, kind,
                         dropout_of=None, n_pos_global=None, n_neg_global=None):
    """``KGEModule.training_step``'s loss (reference kge_module.py:110-129) with the encoder over the rank's shard
    and the scorer over the rank's share of the triplets (``pos_* / neg_*``: global node ids).  The BCE terms are
    normalised by the GLOBAL triplet count over ``world`` (``n_*_global``; default: equal shares), so the bucket's
    average is the whole-graph mean.  The scorer reads the all-gathered embedding table (bf16 on the wire in bf16 mode)
    through slot indices; its gradient comes back through the reduce-scatter."""
    from . import ops
    part = prg.part
    z_local = rgcn_encoder_partitioned(model.encoder, x_local, prg, backend, dropout_of)
    ps = torch.stack([part.slots(pos_ei[0]), part.slots(pos_ei[1])])
    ns = torch.stack([part.slots(neg_ei[0]), part.slots(neg_ei[1])])
    rel = model.decoder.rel_emb
    total = None
    if n_pos_global is not None:
        total = (n_pos_global + n_neg_global) / part.world
    ce, _, _ = ops.kge_bce_loss(z_local, rel, ps, pos_et, ns, neg_et, kind, comm=part.comm, total=total)
    # L2 terms: every rank adds its nodes' share of mean(z^2) (x world for the gradient average) and the full rel term
    reg = z_local.float().pow(2).sum() * (part.world / (part.num_nodes * z_local.size(1))) + rel.pow(2).mean()
    return ce + 1e-2 * reg


# ----------------------------------------------------------------------------- NT-Xent, rows partitioned
class _NtXentPartitioned(torch.autograd.Function):
    """GRACE's loss (reference gcl_module.py:171-173,186-190) with the 2N rows of Z = [z1; z2] partitioned like the
    nodes: the normalised rows are all-gathered, each rank computes D_i for ITS anchor rows against all 2N columns,
    D is all-gathered, and the backward needs nothing else: dL/dz_i = 1/(M tau) (sum_j E_ij (1/D_i + 1/D_j) z_j - 2 z_p(i))
    only involves row i's tiles.  Returns ``world / M * sum_(local i) (log D_i - z_i.z_p(i) / tau)`` -- the rank's share
    scaled so that the gradient AVERAGE over ranks is the whole-graph gradient; ``loss_value`` (all-reduced mean of the
    returns) is the reference's loss."""

    @staticmethod
    def forward(ctx, h1, h2, tau, part, kernels):
        comm, n_loc, n = part.comm, part.n_local, part.num_nodes
        zn_loc, norm, posdot = kernels.normalize(h1, h2)                    # [2 n_loc, hpad], [2 n_loc], [n_loc]
        hpad = zn_loc.size(1)
        # all-gather both views' normalised rows, drop the padding: Z_all [2N, hpad] in (view, node) order
        pair = torch.cat([zn_loc[:n_loc], zn_loc[n_loc:]], dim=1)           # [n_loc, 2 hpad]: one exchange
        table = comm.all_gather(pair)[part.node_slots()]                    # [N, 2 hpad]
        z_all = torch.cat([table[:, :hpad], table[:, hpad:]], dim=0).contiguous()
        lo = part.lo
        anchors = torch.cat([torch.arange(lo, lo + n_loc, device=h1.device),
                             torch.arange(n + lo, n + lo + n_loc, device=h1.device)]).to(torch.int32)
        den_loc = kernels.row_sums(z_all, anchors, tau)                     # [2 n_loc]
        den_all = comm.all_gather_vector(den_loc.view(2, n_loc), fill=1.0)[:, part.node_slots()].reshape(-1).contiguous()
        terms = torch.log(den_loc.double()).sum() - 2.0 * posdot.double().sum() / tau
        ctx.tau, ctx.part, ctx.kernels = tau, part, kernels
        ctx.save_for_backward(z_all, norm, den_all, anchors)
        return (terms * (part.world / (2.0 * n))).float()

    @staticmethod
    def backward(ctx, g):
        z_all, norm, den_all, anchors = ctx.saved_tensors
        part = ctx.part
        n_loc, n = part.n_local, part.num_nodes
        scale = g * (part.world / (2.0 * n * ctx.tau))
        dh = ctx.kernels.row_grads(z_all, anchors, den_all, norm, ctx.tau, scale, n_loc)   # [2 n_loc, h]
        return dh[:n_loc], dh[n_loc:], None, None, None


def ntxent_partitioned(h1_local, h2_local, tau, part: NodePartition, kernels=None):
    """See ``_NtXentPartitioned``.  ``kernels``: the arithmetic (default: the HIP NT-Xent kernels, csrc/ntxent2.hip)."""
    if kernels is None:
        from .ops import NtXentKernels
        kernels = NtXentKernels()
    return _NtXentPartitioned.apply(h1_local, h2_local, float(tau), part, kernels)


def loss_value(loss_local, world):
    """The whole-graph loss from the ranks' scaled shares (reporting only: one scalar all-reduce)."""
    if world == 1:
        return loss_local.detach()
    t = loss_local.detach().clone()
    dist.all_reduce(t)
    return t / world
