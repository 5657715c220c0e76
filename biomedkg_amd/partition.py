"""Node-partitioned full-graph training across the GPUs of one node (BASELINE configs[3]): rank p owns a
contiguous range of nodes -- chosen so that every rank holds about the same number of incident edges -- stores the CSR
rows of its nodes' IN-edges (forward) and OUT-edges (backward) and the activations of its nodes, and keeps a full replica
of the weights.  Nothing here has a reference counterpart: the reference's only multi-GPU mode is Lightning DDP over
mini-batches (``bench.py --mode replicas``).

Exchange (``Comm``): **gathers only, bf16 on the wire, no reduce-scatter anywhere.**

* forward, per conv layer: the rank's rows of the gather table (GCN: after the ``x W^T`` transform, so the wire carries
  ``out`` columns; R-GCN: the layer input) go to the ranks whose in-edges read them;
* backward, per conv layer: the rank's rows of the (gated, bf16) OUTPUT gradient go to the ranks whose out-edges point at
  them, and every rank computes the table gradient of ITS OWN rows from its out-edge CSR:
  ``dXW[j] = sum over out-edges (j -> i) of w_ji dOut[i]`` -- the same sum the single-GPU layer runs, so the partitioned
  backward is the single-GPU arithmetic.  (Round 2 reduce-scattered fp32 partial tables ``A_loc^T dOut`` instead: four
  times the bytes of this bf16 gather, plus an ``[N, d]`` fp32 scratch table per layer and rank.)  R-GCN's table gradient
  is regrouped the same way: ``dX[s] = sum_r (sum over out-edges s -> i of type r of dOut[i] / |N_r(i)|) W_r^T`` -- a
  (relation, source) segment reduce of the gathered ``dOut``, the per-relation products with ``W_r^T``, a per-source sum:
  the forward's three kernels over the reversed local graph;
* ``HaloPattern``: which rows each peer needs is known to every rank (all hold the edge list), so only THOSE rows travel
  (one ``all_to_all_single`` with per-peer counts, packed by a row gather, landed by a row scatter); where nearly all
  rows are needed anyway -- a randomly wired graph -- the pattern falls back to the plain all-gather (no pack / unpack);
* every exchange is begin / end split: R-GCN's root product runs under the input gather, its weight-gradient products
  under the gather of ``dOut``; GCN's product is cut into row chunks whose rows leave while the next chunk is computed.

Weight gradients are averaged with the flat bucket of ``distributed.GradBucket``; losses are scaled by
``world / global count`` so that this AVERAGE is the gradient of the whole-graph loss.  NT-Xent is ROW partitioned
(``ntxent_partitioned``): the normalised rows and the row sums D are all-gathered, each rank runs its anchor rows against
all columns; because W_ij = E_ij (1/D_i + 1/D_j) is symmetric no gradient has to be sent back.

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
class HaloPattern:
    """Who needs which rows.  ``needed_by [world, N]`` (bool, the same on every rank): rank q's CSRs reference node g.
    From it every rank derives its side of the exchange without talking to anybody: the rows it sends to each peer
    (ascending node id) and the table slots where each peer's rows land (the same order) -- whole, and cut into
    ``CHUNKS`` row ranges per owner (chunk c of rank q = its local rows ``[c n_q / CHUNKS, (c + 1) n_q / CHUNKS)``) for
    producers that hand their rows over while still computing the next ones."""

    CHUNKS = 4

    def __init__(self, part, needed_by):
        rank, world, b = part.rank, part.world, part.bounds
        self.world = world
        dev = needed_by.device
        k = self.CHUNKS
        cut = lambda q, c: b[q] + ((b[q + 1] - b[q]) * c) // k                                     # noqa: E731
        self.chunk_rows = [(cut(rank, c) - b[rank], cut(rank, c + 1) - b[rank]) for c in range(k)]  # my local row ranges
        # every rank reads every row ("all" of unequal shards): each peer's rows are one contiguous run of the table's slot
        # space and mine one contiguous run of my shard -- the exchange can land them in place (Comm.exchange_begin)
        self.whole = bool(needed_by.all())
        self.peer_chunk_rows = [[(cut(q, c) - b[q], cut(q, c + 1) - b[q]) for q in range(world)] for c in range(k)]
        self.peer_rows = [(0, b[q + 1] - b[q]) for q in range(world)]
        self.parts = []                      # per chunk: (send_rows, send_counts, recv_slots, recv_counts)
        for c in range(k):
            send, recv, sc, rc = [], [], [], []
            for q in range(world):
                if q == rank:
                    sc.append(0); rc.append(0)
                    continue
                lo, hi = cut(rank, c), cut(rank, c + 1)
                mine = torch.nonzero(needed_by[q, lo:hi]).squeeze(1) + (lo - b[rank])            # my rows q reads
                lo, hi = cut(q, c), cut(q, c + 1)
                theirs = torch.nonzero(needed_by[rank, lo:hi]).squeeze(1) + (lo - b[q]) + q * part.n_max   # slots
                send.append(mine); recv.append(theirs)
                sc.append(int(mine.numel())); rc.append(int(theirs.numel()))
            cat = lambda parts: (torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=dev)).contiguous()  # noqa: E731
            self.parts.append((cat(send), sc, cat(recv), rc))
        # the whole exchange at once: per peer, the chunks' lists one after the other
        self.send_counts = [sum(p[1][q] for p in self.parts) for q in range(world)]
        self.recv_counts = [sum(p[3][q] for p in self.parts) for q in range(world)]

        def whole(idx, cnt):
            out = []
            for q in range(world):
                for c in range(k):
                    off = sum(self.parts[c][cnt][:q])
                    out.append(self.parts[c][idx][off:off + self.parts[c][cnt][q]])
            return torch.cat(out).contiguous() if out else torch.zeros(0, dtype=torch.int64, device=dev)
        self.send_rows, self.recv_slots = whole(0, 1), whole(2, 3)
        self.n_send, self.n_recv = int(self.send_rows.numel()), int(self.recv_slots.numel())
        # a collective decision: the halo exchange only if EVERY rank reads a clearly smaller part of the others' rows
        share = []
        for q in range(world):
            own = int(needed_by[q, b[q]:b[q + 1]].sum())
            share.append((int(needed_by[q].sum()) - own) / max(part.num_nodes - (b[q + 1] - b[q]), 1))
        self.max_fraction = max(share) if world > 1 else 0.0


HALO_THRESHOLD = float(os.environ.get("BKG_HALO_THRESHOLD", "0.75"))   # halo exchange when every rank reads <= this share
EXACT_GATHER_WASTE = float(os.environ.get("BKG_EXACT_GATHER_WASTE", "1.1"))   # padded / real rows above which "all" is exact


class _Done:
    """A collective that has already happened (``Comm(simulate=True)``)."""

    @staticmethod
    def wait():
        return True


class Comm:
    """The exchange of one partition: ``exchange_begin`` / ``exchange_end`` move the rank's shard of a node table to the
    ranks that read it (RCCL / gloo run the collective on their own stream; ``end`` makes the current stream wait for
    it) and return the ``[world * n_max, d]`` table in slot space.  ``stats`` counts what crossed the wire and, with
    ``timed``, how long the compute stream waited (``exposed``) -- bench.py's ``wire_bytes_per_step`` /
    ``exchange_exposed_ms``.

    ``simulate=True`` (one process standing in for rank ``rank`` of ``world``; bench.py's simulated-rank measurement): no
    process group -- every exchange packs and lands exactly as the live one does, but the rows that would arrive over the
    wire are a LOCAL copy of the same byte count (own rows, repeated), so the rank's kernels, pack / unpack passes and
    buffer sizes are those of the real rank and only the wire itself is missing."""

    def __init__(self, world, n_local, n_max, group=None, rank=0, simulate=False):
        self.world, self.n_local, self.n_max, self.group, self.rank = int(world), int(n_local), int(n_max), group, int(rank)
        self.simulate = bool(simulate)
        # one rank has nothing to exchange; BKG_PARTITION_FORCE_COLLECTIVES=1 issues the collectives anyway (a 1-rank
        # RCCL group: the way to run the exact collective calls of the N-rank path on a single-GPU box)
        self.live = self.world > 1 or os.environ.get("BKG_PARTITION_FORCE_COLLECTIVES") == "1"
        self.patterns = {}          # name -> HaloPattern (None: every row travels -- the plain all-gather)
        self.serial = os.environ.get("BKG_PARTITION_SERIAL") == "1"     # begin() waits at once: nothing is hidden
        self.timed = False
        self.stats = {"bytes_received": 0, "bytes_sent": 0, "exchanges": 0}
        self._waits = []            # (event before the wait, event after it) of this step

    # ---- buffers
    def table_buffer(self, d, dtype, device):
        """The full ``[world * n_max, d]`` table for a producer to write its rows straight into (``own_rows``): the
        exchange then fills in the other ranks' rows around them -- no copy of the shard."""
        return torch.empty(self.world * self.n_max, d, dtype=dtype, device=device)

    def own_rows(self, table):
        """The rank's ``n_max`` padded rows of a table (the pad rows are never referenced by any CSR)."""
        return table[self.rank * self.n_max:(self.rank + 1) * self.n_max]

    def shard_buffer(self, d, dtype, device):
        """A padded ``[n_max, d]`` shard that IS the rank's slice of a fresh table (``exchange`` recognises it)."""
        table = self.table_buffer(d, dtype, device)
        shard = self.own_rows(table)
        shard._bkg_table = table
        return shard

    def _table_of(self, shard):
        table = getattr(shard, "_bkg_table", None)
        if table is not None and table.size(1) == shard.size(1) and table.dtype == shard.dtype:
            return table
        if not self.live:                                  # one rank: the (padded) shard IS the table
            if shard.size(0) == self.n_max:
                return shard if shard.is_contiguous() else shard.contiguous()
            table = torch.empty(self.n_max, shard.size(1), dtype=shard.dtype, device=shard.device)
            table[: shard.size(0)].copy_(shard)
            return table
        table = self.table_buffer(shard.size(1), shard.dtype, shard.device)
        self.own_rows(table)[: shard.size(0)].copy_(shard)
        return table

    def chunks(self, pattern):
        """Row ranges ``[(lo, hi), ...]`` of the rank's shard that ``exchange_begin(.., chunk=c)`` sends one by one; one
        range (everything) where this pattern moves whole shards."""
        pat = self.patterns.get(pattern) if self.live else None
        return pat.chunk_rows if pat is not None else [(0, self.n_local)]

    # ---- the exchange
    def exchange_begin(self, shard, pattern="in", chunk=None):
        """Start moving the rank's rows (``shard [n_local | n_max, d]``) to their readers.  ``chunk``: only that row
        range of ``chunks(pattern)`` leaves now (a product still being computed hands over what is finished); the table
        is complete once every chunk's ``exchange_end`` has run."""
        table = self._table_of(shard)
        if not self.live:
            return table, None
        pat = self.patterns.get(pattern)
        el = table.element_size() * table.size(1)
        own = self.own_rows(table)
        if pat is None:
            # every row travels: the plain all-gather (from a copy of the shard: it already sits at its offset of the
            # table, and an input that aliases the output is not something every backend promises to take)
            if self.simulate:
                table.view(self.world, self.n_max, table.size(1)).copy_(own.clone().unsqueeze(0).expand(self.world, -1, -1))
                work = _Done
            else:
                work = dist.all_gather_into_tensor(table, own.clone(), group=self.group, async_op=True)
            self._count(el * self.n_max * (self.world - 1), el * self.n_max * (self.world - 1))
            return self._issued((table, work, None, None))
        if pat.whole and self._lands_in_place():
            # every row to every rank, unequal shards: my rows leave from where they are and each peer's rows arrive where
            # they belong (RCCL runs the list form of all-to-all as grouped sends / receives on the given tensors) -- no
            # (world - 1)-fold packed copy of the shard, no unpack pass over the other ranks' rows
            lo, hi = (0, self.n_local) if chunk is None else pat.chunk_rows[chunk]
            ranges = pat.peer_rows if chunk is None else pat.peer_chunk_rows[chunk]
            send = own[lo:hi]
            outs = [table[q * self.n_max + a:q * self.n_max + (b if q != self.rank else a)] for q, (a, b) in enumerate(ranges)]
            rows_in = sum(o.size(0) for o in outs)
            if self.simulate:
                for o in outs:
                    if o.size(0):
                        o.copy_(own.index_select(0, self._stand_in_rows(o.size(0), own.device)))
                work = _Done
            else:
                ins = [send if q != self.rank else send[:0] for q in range(self.world)]
                work = dist.all_to_all(outs, ins, group=self.group, async_op=True)
            self._count(el * rows_in, el * send.size(0) * (self.world - 1))
            return self._issued((table, work, None, None))
        if chunk is None:
            send_rows, send_counts, recv_slots, recv_counts = pat.send_rows, pat.send_counts, pat.recv_slots, pat.recv_counts
        else:
            send_rows, send_counts, recv_slots, recv_counts = pat.parts[chunk]
        packed = own.index_select(0, send_rows)
        if self.simulate:
            recv, work = own.index_select(0, self._stand_in_rows(sum(recv_counts), own.device)), _Done
        else:
            recv = torch.empty(sum(recv_counts), table.size(1), dtype=table.dtype, device=table.device)
            work = dist.all_to_all_single(recv, packed, recv_counts, send_counts, group=self.group, async_op=True)
        self._count(el * recv.size(0), el * packed.size(0))
        return self._issued((table, work, recv, recv_slots))

    def _lands_in_place(self):
        """The list form of all-to-all on the tensors themselves -- my rows leave from their slice of the table, each peer's
        rows arrive in theirs (disjoint row ranges of one allocation) -- is OPT-IN: ``BKG_DIRECT_ALL_TO_ALL=1``.
        It has only ever run in the simulated rank (it needs RCCL ranks with unequal shards; no multi-GPU box was available
        to this build), so library users under torchrun get the packed ``all_to_all_single`` -- the form every gloo world of
        tests/test_partition_gloo.py runs -- unless they ask.  bench.py asks for it in its first, time-fenced attempts and
        falls back to the packed form (``run_partition_ladder``).  gloo has no list form at all."""
        if os.environ.get("BKG_DIRECT_ALL_TO_ALL", "0") != "1":
            return False
        return self.simulate or (dist.is_initialized() and dist.get_backend(self.group) == "nccl")

    def _stand_in_rows(self, count, device):
        """(simulate) ``count`` row numbers of the rank's own shard, cycling: what stands in for the rows off the wire."""
        hit = getattr(self, "_stand_in", None)
        if hit is None or hit.numel() < count or hit.device != device:
            hit = torch.arange(max(count, 1), device=device) % max(self.n_local, 1)
            self._stand_in = hit
        return hit[:count]

    def _issued(self, pending):
        if self.serial:                                    # (measurement: the whole collective shows as exposed time)
            return self.exchange_end(pending), None
        return pending

    def exchange_end(self, pending):
        if pending[1] is None:
            return pending[0]
        table, work, recv, recv_slots = pending
        if self.timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        work.wait()
        if self.timed:
            b.record()
            self._waits.append((a, b))
        if recv is not None and recv.size(0):
            table.index_copy_(0, recv_slots, recv)
        return table

    def exchange(self, shard, pattern="in"):
        return self.exchange_end(self.exchange_begin(shard, pattern))

    # ---- tables of constant inputs (the first R-GCN layer gathers the node features: the same rows every step)
    def constant_table(self, holder, shard, pattern="in"):
        """The gathered table of ``shard`` remembered on ``holder`` (a graph structure) under ``pattern``, or None.  Valid
        for this tensor object at this version; the holder keeps the shard alive, so its address cannot be handed to
        another tensor."""
        hit = (getattr(holder, "_bkg_const_tables", None) or {}).get(pattern)
        if hit is not None and hit[0] is shard and hit[1] == shard._version and hit[2].device == shard.device:
            return hit[2]
        return None

    @staticmethod
    def is_constant_input(needs_input_grad, training_pass=True):
        """May the layer input's gathered table be remembered?  Only in a TRAINING pass (``training_pass``: grad mode as
        the layer's caller had it -- inside an autograd Function's forward it always reads off) whose input is a constant:
        the input wants no gradient while some parameter does.  Under ``torch.no_grad()`` (an eval pass, bench.py's check
        losses) EVERY hidden activation looks constant: remembering those pinned a ``[world * n_max, d]`` table per layer
        on the graph and evicted the features' table on each layer.  (A table remembered by a training pass is
        still USED by a later no-grad pass over the same tensor: identity and version decide that.)"""
        return bool(training_pass) and (not needs_input_grad[0]) and any(needs_input_grad[1:])

    def remember_table(self, holder, shard, pattern, table):
        """One table per pattern key on the holder (the composable form and the fused layers use different keys: they do
        not evict each other)."""
        tables = getattr(holder, "_bkg_const_tables", None)
        if tables is None:
            tables = holder._bkg_const_tables = {}
        tables[pattern] = (shard, shard._version, table)
        while len(tables) > 4:                           # a handful of (pattern, precision) keys at most
            tables.pop(next(iter(tables)))

    def _count(self, received, sent):
        self.stats["bytes_received"] += int(received)
        self.stats["bytes_sent"] += int(sent)
        self.stats["exchanges"] += 1

    def take_stats(self):
        """-> (bytes received, bytes sent, exchanges, exposed ms) since the last call (``timed``: synchronises)."""
        exposed = None
        if self.timed:
            torch.cuda.synchronize()
            exposed = sum(a.elapsed_time(b) for a, b in self._waits)
        out = (self.stats["bytes_received"], self.stats["bytes_sent"], self.stats["exchanges"], exposed)
        self.stats = {"bytes_received": 0, "bytes_sent": 0, "exchanges": 0}
        self._waits = []
        return out

    # ---- whole-table exchange (once per step: the scorer's embedding table, NT-Xent's rows, GGD's shared first product)
    def all_gather(self, shard):
        """Every row to every rank: the plain all-gather for equal shards, exact per-peer counts otherwise (pattern "all")."""
        return self.exchange(shard, pattern="all")

    def all_gather_vector(self, v_local, fill=0.0):
        """``[n_local]`` (or ``[k, n_local]``) per rank -> ``[k, world * n_max]`` in slot order."""
        v = v_local.reshape(-1, v_local.size(-1))
        buf = torch.full((v.size(0), self.n_max), fill, dtype=v.dtype, device=v.device)
        buf[:, : v.size(1)] = v
        if not self.live:
            return buf
        if self.simulate:
            out = buf.repeat(self.world, 1)
        else:
            out = torch.empty(self.world * v.size(0), self.n_max, dtype=v.dtype, device=v.device)
            dist.all_gather_into_tensor(out, buf, group=self.group)
        return out.view(self.world, v.size(0), self.n_max).permute(1, 0, 2).reshape(v.size(0), self.world * self.n_max)


# ----------------------------------------------------------------------------- partition
class NodePartition:
    """Contiguous node ranges balanced by ``(in-degree + out-degree) / 2 + node_weight``: the forward's gather /
    aggregate kernels cost per in-edge, the backward's per out-edge, the products, epilogues and row stores per node
    (about 16 edge-gathers' worth at d = 256), and the floor keeps the ranks that own hub nodes from shrinking to a few
    hundred rows.  Holds the two halo patterns of the edge set (``comm.patterns``): ``"in"`` -- the rows the rank's
    in-edges read (forward tables) -- and ``"out"`` -- the rows its out-edges point at (the gathered ``dOut``)."""

    NODE_WEIGHT = 16.0

    def __init__(self, edge_index, num_nodes, rank, world, node_weight=None, halo=None, simulate=False):
        self.num_nodes, self.rank, self.world = int(num_nodes), int(rank), int(world)
        dev = edge_index.device
        self.node_weight = float(self.NODE_WEIGHT if node_weight is None else node_weight)
        load = self.load_of(edge_index, num_nodes, self.node_weight)
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
        self.comm = Comm(world, self.n_local, self.n_max, rank=rank, simulate=simulate)
        self._valid = None
        self.halo_fraction = {}
        self.padding_waste = 1.0
        if world > 1:
            # whole-table exchanges (the scorer's embedding table, GGD's paired first product, NT-Xent's rows): the padded
            # all-gather moves (world - 1) n_max rows to every rank whatever its peers really own -- on the bench graph at
            # 8 ranks 204,638 rows where the other ranks hold 100,766..125,908 (load-balanced shards of 4,092..29,234
            # rows).  When that padding is worth more than 10 % somewhere, "every row" becomes a pattern of its own
            # (exact per-peer counts through the same all-to-all as the halo rows); equal shards keep the all-gather.
            sizes = [b - a for a, b in zip(bounds[:-1], bounds[1:])]
            self.padding_waste = (world - 1) * self.n_max / max(self.num_nodes - max(sizes), 1)
            exact = self.padding_waste > EXACT_GATHER_WASTE if halo is not False else False
            everything = None
            if exact:
                everything = HaloPattern(self, torch.ones(world, self.num_nodes, dtype=torch.bool, device=dev))
            self.comm.patterns["all"] = everything
            src, dst = edge_index[0], edge_index[1]
            own_s = torch.bucketize(src, self._bounds_t[1:-1], right=True)
            own_d = torch.bucketize(dst, self._bounds_t[1:-1], right=True)
            for name, reader, read in (("in", own_d, src), ("out", own_s, dst)):
                needed = torch.zeros(world, self.num_nodes, dtype=torch.bool, device=dev)
                needed[reader, read] = True                       # rank `reader` reads node `read`'s row
                pat = HaloPattern(self, needed)
                self.halo_fraction[name] = pat.max_fraction
                # every rank takes the same decision (max_fraction is a global quantity): halo rows only when that saves
                # a clear share of the table, else every row (no pack / unpack when the shards are equal: the plain
                # all-gather); halo=True / False forces it
                use = pat.max_fraction <= HALO_THRESHOLD if halo is None else bool(halo)
                self.comm.patterns[name] = pat if use else everything

    @staticmethod
    def load_of(edge_index, num_nodes, node_weight):
        deg = torch.bincount(edge_index[1], minlength=num_nodes) + torch.bincount(edge_index[0], minlength=num_nodes)
        return deg.double() * 0.5 + node_weight

    def slots(self, ids):
        """global node ids -> rows of the gathered ``[world * n_max, d]`` table"""
        if self.world == 1:
            return ids
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
    """The rank's slices of ``A_hat = D^-1/2 (A + I) D^-1/2`` (PyG gcn_norm semantics) with the unit self
    loops as explicit entries: ``by_dst`` -- rows = its nodes as TARGETS, columns = table slots of the sources (forward:
    ``out = by_dst @ gathered table``) -- and ``by_src`` -- rows = its nodes as SOURCES, columns = table slots of the
    targets (backward: ``dXW = by_src @ gathered dOut``: every rank sums the table gradient of its own rows, nothing is
    reduce-scattered).  Carries what the fused layers read off a graph (``by_dst`` / ``by_src`` / ``n_out`` / ``n_table``
    / ``comm``); ``gcn_weights(keep)`` re-weights the same CSRs for a ``dropout_edge`` view -- global degrees from the full
    edge list, no communication, nothing re-sorted -- and, on the product backend for a shard of ``KEPT_CSR_MIN_EDGES``
    entries or more, cuts the view's own CSRs out of them (``graph.KeptCSR.of_weights``: the dropped edges' rows are then
    never gathered, as in the single-GPU ``Graph.gcn_view``)."""

    def __init__(self, part: NodePartition, edge_index, backend, keep=None):
        self.part, self.backend, n = part, backend, part.num_nodes
        self.edge_index = edge_index
        src, dst = edge_index[0], edge_index[1]
        e = edge_index.size(1)
        own = torch.arange(part.lo, part.hi, device=edge_index.device)
        own_slot = part.slots(own)
        eid = torch.nonzero(part.is_local(dst)).squeeze(1)                  # in-edges of my nodes
        self._entry_in = torch.cat([eid, e + own])                          # entry -> index into cat(w_edge, self_w)
        self.by_dst = backend.csr(torch.cat([dst[eid] - part.lo, own - part.lo]),      # rows: local targets
                                  torch.cat([part.slots(src[eid]), own_slot]), part.n_local)   # neighbours + the self loop
        eid = torch.nonzero(part.is_local(src)).squeeze(1)                  # out-edges of my nodes
        self._entry_out = torch.cat([eid, e + own])
        self.by_src = backend.csr(torch.cat([src[eid] - part.lo, own - part.lo]),      # rows: local sources
                                  torch.cat([part.slots(dst[eid]), own_slot]), part.n_local)
        self.num_nodes = self.n_out = part.n_local
        self.n_table = part.world * part.n_max
        self.comm = part.comm
        self._weights = {}
        self._default_keep = keep
        gw = self.gcn_weights(keep)
        self.fwd, self.bwd = gw.dst or self.by_dst, gw.src or self.by_src      # (the composable form's pair: CSR + its weights)
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
        w = torch.cat([w_edge, self_w])
        w_dst = self.backend.permute(self.by_dst, w[self._entry_in])
        w_src = self.backend.permute(self.by_src, w[self._entry_out])
        kept = getattr(self.backend, "kept_csr", None)
        if keep is not None and kept is not None and min(self.by_dst.nnz, self.by_src.nnz) >= _kept_floor():
            kd, ks = kept(self.by_dst, w_dst), kept(self.by_src, w_src)
            out = GcnWeights(kd.w, ks.w, None, dst=kd, src=ks)
        else:
            out = GcnWeights(w_dst, w_src, None)
        self._weights["kept" if keep is not None else "plain"] = (key, out, keep)   # (keep kept alive: id() stays unique)
        return out


def _kept_floor():
    from . import graph
    return graph.KEPT_CSR_MIN_EDGES


class _PartSpmm(torch.autograd.Function):
    """``epi(fwd @ gather(x_local))`` over a partition (composable form; the product's fused layers do the same inside
    ``ops``): forward -- the rank's rows go to their readers (pattern "in"), the rank reduces its target rows; backward --
    the (gated) output gradient goes to ITS readers (pattern "out") and the rank reduces the table gradient of its own
    rows over its out-edge CSR.  No reduce-scatter."""

    @staticmethod
    def forward(ctx, x_local, addend, bias, comm, backend, fwd, bwd, w_fwd, w_bwd, relu, drop):
        table = comm.exchange(x_local, "in")
        out = backend.spmm_epilogue(fwd, w_fwd, table, fwd.n_rows, addend, bias, relu, drop)
        ctx.comm, ctx.backend, ctx.bwd, ctx.w_bwd, ctx.relu, ctx.drop = comm, backend, bwd, w_bwd, relu, drop
        ctx.n_local, ctx.x_dtype = x_local.size(0), x_local.dtype
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dout = ctx.backend.gate(dout, out, ctx.relu, ctx.drop)
        need = ctx.needs_input_grad
        dx = None
        if need[0]:
            dtable = ctx.comm.exchange(dout.contiguous(), "out")
            dx = ctx.backend.spmm_epilogue(ctx.bwd, ctx.w_bwd, dtable, ctx.bwd.n_rows, None, None, False, None)
            dx = dx[: ctx.n_local].to(ctx.x_dtype)
        return (dx, dout if need[1] else None, dout.sum(0) if need[2] else None, None, None, None, None, None, None, None,
                None)


def part_spmm(x_local, comm, backend, fwd, bwd, w_fwd, w_bwd, addend=None, bias=None, relu=False, drop=None):
    return _PartSpmm.apply(x_local, addend, bias, comm, backend, fwd, bwd, w_fwd, w_bwd, relu, drop)


def gcn_conv_partitioned(x_local, weight, bias, pg: PartitionedGraph, backend, relu=False, drop=None, gw=None):
    """``dropout(relu(A_hat[local rows] @ gather(x_local W^T) + b))`` -> ``[n_local, out]`` (composable form)."""
    gw = gw if gw is not None else pg.gcn_weights()
    xw = backend.linear(x_local, weight)                                     # transform BEFORE the exchange:
    return part_spmm(xw, pg.comm, backend, gw.dst or pg.by_dst, gw.src or pg.by_src, gw.w_dst, gw.w_src, bias=bias, relu=relu,
                     drop=drop)


class _RelSide:
    """(relation, local node) segments over table slots for one direction of the R-GCN layer (composable form):
    ``seg`` -- rows = segments, columns = table slots, per-entry weights ``seg_w``; ``rows`` -- rows = local nodes,
    columns = their segments (plain sum); ``rel_ptr`` -- segment ranges per relation; ``seg_node`` -- each segment's
    local node."""

    def __init__(self, backend, node_local, slot_other, edge_type, w_edge, n_local, num_relations):
        key = edge_type * max(n_local, 1) + node_local                       # sorted by (relation, local node)
        uniq, seg_of_edge = torch.unique(key, return_inverse=True)
        self.num_segments = int(uniq.numel())
        self.seg_node = uniq % max(n_local, 1)
        seg_rel = uniq // max(n_local, 1)
        self.seg = backend.csr(seg_of_edge, slot_other, self.num_segments)
        self.seg_w = backend.permute(self.seg, w_edge)
        iota = torch.arange(self.num_segments, device=node_local.device)
        self.rows = backend.csr(self.seg_node, iota, n_local)
        self.rel_ptr = torch.searchsorted(seg_rel, torch.arange(num_relations + 1, device=seg_rel.device)).to(torch.int32)


class PartitionedRelGraph:
    """The rank's slices of the R-GCN mean aggregation (PyG RGCNConv aggr="mean").  Forward side: the
    (relation, local TARGET) segments of the rank's in-edges over the table slots of their sources, mean weights
    ``1 / |N_r(i)|``.  Backward side: the (relation, local SOURCE) segments of its out-edges over the slots of their
    targets, each edge carrying the SAME weight it has in the forward (``1 / |N_r(target)|``, a global count every rank
    derives from the full edge list) -- ``dX[s] = sum_r (sum_e w_e dOut[target_e]) W_r^T`` runs the forward's three
    kernels over it, so no partial table is ever reduce-scattered.  With ``HipBackend`` both sides are ``graph.RelGraph``
    objects (``fused`` / ``fused.rev``), i.e. what ``ops.rgcn_conv`` takes."""

    def __init__(self, part: NodePartition, edge_index, edge_type, num_relations, backend):
        self.part, self.num_relations = part, int(num_relations)
        src, dst = edge_index[0], edge_index[1]
        n = part.num_nodes
        # 1 / |N_r(i)| of every edge: in-edges of type r at its target (global: all edges are known to every rank)
        _, inv, cnt = torch.unique(edge_type * n + dst, return_inverse=True, return_counts=True)
        w_edge = 1.0 / cnt[inv].to(torch.float32)
        mine = part.is_local(dst)
        outs = part.is_local(src)
        self.n_table = part.world * part.n_max
        fwd_edges = torch.stack([part.slots(src[mine]), dst[mine] - part.lo])
        rev_edges = torch.stack([part.slots(dst[outs]), src[outs] - part.lo])
        self.fused = backend.rel_graph(fwd_edges, edge_type[mine], part.n_local, self.num_relations, self.n_table, part.comm,
                                       rev=(rev_edges, edge_type[outs], w_edge[outs]))
        if self.fused is not None:
            return
        self.fwd = _RelSide(backend, fwd_edges[1], fwd_edges[0], edge_type[mine], w_edge[mine], part.n_local, self.num_relations)
        self.bwd = _RelSide(backend, rev_edges[1], rev_edges[0], edge_type[outs], w_edge[outs], part.n_local, self.num_relations)


class _PartRgcn(torch.autograd.Function):
    """One RGCNConv over the rank's nodes (composable form; ``ops._RgcnConv`` is the product's fused one): forward --
    gather the layer input (pattern "in"), segment means, per-relation products, per-target sum + root + bias; backward
    -- local weight / root / bias gradients, then gather ``dOut`` (pattern "out") and regroup: (relation, source)
    segment sums of it, products with ``W_r^T``, per-source sum + root part."""

    @staticmethod
    def forward(ctx, x_local, weight, root, bias, prg, backend, relu, drop, training_pass=True):
        comm, f = prg.part.comm, prg.fwd
        constant = comm.is_constant_input(ctx.needs_input_grad, training_pass)    # the node features: the same rows every step, gathered once
        table = None if ctx.needs_input_grad[0] else comm.constant_table(prg, x_local)
        if table is None:
            table = comm.exchange(x_local, "in")
            if constant:
                comm.remember_table(prg, x_local, "in", table)
        h = backend.spmm_epilogue(f.seg, f.seg_w, table, f.num_segments, None, None, False, None)
        y = backend.grouped_linear(h, weight, f.rel_ptr)
        out = backend.spmm_epilogue(f.rows, None, y, x_local.size(0), backend.matmul(x_local, root), bias, relu, drop)
        ctx.prg, ctx.backend, ctx.relu, ctx.drop = prg, backend, relu, drop
        ctx.save_for_backward(x_local, weight, root, h, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x_local, weight, root, h, out = ctx.saved_tensors
        prg, be = ctx.prg, ctx.backend
        comm, f, b = prg.part.comm, prg.fwd, prg.bwd
        dout = be.gate(dout, out, ctx.relu, ctx.drop).contiguous()
        need_dx = ctx.needs_input_grad[0]
        pending = comm.exchange_begin(dout, "out") if need_dx else None      # crosses the wire under the local products
        dweight = be.grouped_outer(h, dout[f.seg_node], f.rel_ptr, weight.size(0))        # dW_r = H_r^T dY_r
        droot = be.matmul_tn(x_local, dout)                                  # x^T dOut, through the backend like every product
        db = dout.sum(0)
        dx = None
        if need_dx:
            dtable = comm.exchange_end(pending)
            t = be.spmm_epilogue(b.seg, b.seg_w, dtable, b.num_segments, None, None, False, None)
            dxs = be.grouped_linear(t, weight.transpose(1, 2), b.rel_ptr)
            dx = be.spmm_epilogue(b.rows, None, dxs, x_local.size(0), be.matmul_nt(dout, root), None, False, None)
        return dx, dweight, droot, db, None, None, None, None, None


def rgcn_conv_partitioned(x_local, weight, root, bias, prg: PartitionedRelGraph, backend, relu=False, drop=None):
    """RGCNConv over the rank's targets (composable form)."""
    return _PartRgcn.apply(x_local, weight, root, bias, prg, backend, relu, drop, torch.is_grad_enabled())


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
    def kept_csr(csr, w):
        from .graph import KeptCSR
        return KeptCSR.of_weights(csr, w)

    @staticmethod
    def gcn_norm(edge_index, num_nodes, keep):
        from .graph import graph_of
        return graph_of(edge_index, num_nodes).gcn_norm_raw(keep)

    @staticmethod
    def rel_graph(slot_edges, edge_type, n_local, num_relations, n_table, comm, rev=None):
        from .graph import RelGraph
        live = rev is not None and comm is not None and comm.live      # one rank: the ordinary single-GPU backward
        rg = RelGraph(slot_edges.contiguous(), edge_type.contiguous(), n_local, num_relations, n_src=n_table,
                      defer_source_side=live)
        rg.n_out, rg.n_table, rg.comm = n_local, n_table, comm
        if live:                 # the backward side: (relation, local source) segments over the targets' slots, forward weights
            r_edges, r_type, r_w = rev
            rg.rev = RelGraph(r_edges.contiguous(), r_type.contiguous(), n_local, num_relations, n_src=n_table,
                              defer_source_side=True, edge_weight=r_w.to(torch.float32).contiguous())
        return rg

    @staticmethod
    def linear(x, weight, bias=None):
        from . import ops
        return ops.linear(x, weight, bias)

    @staticmethod
    def matmul(x, w):
        from . import ops
        return ops.gemm(x, w)

    @staticmethod
    def matmul_tn(a, b):
        """``a^T b`` (weight-gradient shape: the rows are contracted)"""
        from . import ops
        return ops.gemm(ops._as_f32(a), ops._as_f32(b), trans_a=True)

    @staticmethod
    def matmul_nt(a, b):
        """``a b^T``"""
        from . import ops
        return ops.gemm(ops._as_f32(a), ops._as_f32(b), trans_b=True)

    @staticmethod
    def gcn_conv(x_local, weight, bias, pg, gw, relu=False, drop=None):
        from . import ops
        return ops.gcn_conv(x_local, weight, bias, pg, gw, relu=relu, drop=drop)

    @staticmethod
    def table_conv(table, bias, pg, gw, relu=False, drop=None):
        return _TableConv.apply(table, bias, pg.comm, HipBackend, gw.dst or pg.by_dst, gw.src or pg.by_src, gw.w_dst, gw.w_src,
                                relu, drop)

    @staticmethod
    def spmm_epilogue(csr, w, table, n_out, addend, bias, relu, drop):
        from . import ops
        return ops.spmm(csr.rowptr, csr.col, w, ops._table(table), n_out, addend=addend, bias=bias, relu=relu, drop=drop,
                        split=csr.split)

    @staticmethod
    def gate(dout, out, relu, drop):
        from . import ops
        scale = 1.0 / (1.0 - drop.p) if (drop is not None and drop.p > 0) else 1.0
        if not relu and scale == 1.0:
            return dout
        if not relu:
            raise ValueError("dropout without a ReLU in front is not on the reference's path")
        return ops._gate(ops._as_f32(dout), ops._as_f32(out), scale)


class _TableConv(torch.autograd.Function):
    """``epi(by_dst @ table)`` for a table that is ALREADY gathered (GGD's paired first product): the gradient handed
    back has the rank's own slots filled (``by_src @ gathered dOut``) and zeros elsewhere -- ``_PairedTable`` knows."""

    @staticmethod
    def forward(ctx, table, bias, comm, backend, fwd, bwd, w_fwd, w_bwd, relu, drop):
        out = backend.spmm_epilogue(fwd, w_fwd, table, fwd.n_rows, None, bias, relu, drop)
        ctx.comm, ctx.backend, ctx.bwd, ctx.w_bwd, ctx.relu, ctx.drop, ctx.shape = comm, backend, bwd, w_bwd, relu, drop, table.shape
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        comm = ctx.comm
        dout = ctx.backend.gate(dout, out, ctx.relu, ctx.drop).contiguous()
        dtable = comm.exchange(dout, "out")
        own = ctx.backend.spmm_epilogue(ctx.bwd, ctx.w_bwd, dtable, ctx.bwd.n_rows, None, None, False, None)
        g = own.new_zeros(ctx.shape)
        (comm.own_rows(g) if comm.live else g)[: own.size(0)] = own
        return g, dout.sum(0), None, None, None, None, None, None, None, None


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


class _PairedTable(torch.autograd.Function):
    """GGD's first layer over a partition.  ``x[perm] W^T = (x W^T)[perm]``: the corrupted view needs ARBITRARY rows of
    the first product, so that one table is all-gathered whole (h columns, once for both views) and paired on every
    rank: slot j -> ``[xw_j | xw_perm(j)]``.  Backward: the consumer hands back the table gradient of the rank's OWN
    slots (both halves; ``ops.gcn_conv_views(is_table=True)`` / ``part_spmm`` compute nothing else); the corrupted
    halves belong to other nodes' rows -- ``d xw[k] = dT_pos[k] + dT_neg[perm^-1(k)]`` -- so they are all-gathered (h
    columns, bf16 in bf16 mode) and every rank picks the rows of its own nodes.  No reduce-scatter."""

    @staticmethod
    def forward(ctx, xw_local, perm_slots, comm, pair):
        table = comm.all_gather(xw_local)
        ctx.comm, ctx.n_local, ctx.h = comm, xw_local.size(0), xw_local.size(1)
        ctx.save_for_backward(perm_slots)
        return pair(table, perm_slots)

    @staticmethod
    def backward(ctx, g):
        (perm_slots,) = ctx.saved_tensors
        comm, h = ctx.comm, ctx.h
        own = comm.own_rows(g) if comm.live else g                          # [n_max, 2h]: the only rows that are filled
        neg_all = comm.all_gather(own[:, h:].contiguous())                  # [n_table, h]
        inv = torch.empty_like(perm_slots)
        inv[perm_slots] = torch.arange(perm_slots.numel(), device=perm_slots.device)
        mine = comm.own_rows(inv) if comm.live else inv                     # perm^-1 of the rank's own slots
        d = own[:, :h].float() + neg_all[mine].float()
        return d[: ctx.n_local].to(g.dtype), None, None, None


def ggd_loss_partitioned(ggd, x_local, pg: PartitionedGraph, backend, perm, dropout_of=None, keep=None, mask_feature=None):
    """``GGDModule.calculate_loss`` (reference gcl_module.py:229-234, gcl.py:73-93) for the rank's nodes: both encoder
    passes run over the rank's shard, the BCE terms of the local nodes are summed and divided by the GLOBAL count, times
    ``world`` so that the bucket's gradient AVERAGE equals the gradient of the whole-graph loss.  ``x_local``: the rank's
    rows of the (possibly feature-masked) input; ``perm``: the corruption permutation of all N nodes and ``keep``: the
    optional ``dropout_edge`` mask over all E edges (every rank must pass the same ones).  The corrupted pass never
    needs other ranks' input rows: ``x[perm] W^T = (x W^T)[perm]``, so the permutation is applied to the gathered table
    of the FIRST layer's product (``_PairedTable``: the wire carries h columns once for both views)."""
    part = pg.part
    total = 2.0 * part.num_nodes
    enc = ggd.encoder
    gw = pg.gcn_weights(keep)
    perm_slots = torch.arange(pg.n_table, dtype=torch.int64, device=perm.device)   # padding slots map to themselves
    node_slots = part.node_slots()
    perm_slots[node_slots] = node_slots[perm]                               # slot(j) -> slot(perm[j]): a bijection
    layer0 = enc.graph_layers[0]
    last = len(enc.graph_layers) - 1
    n_loc = part.n_local
    if backend.fused:
        from . import ops
        n_dec = part.n_max                              # the path is chosen from rank-invariant sizes: all ranks alike
        if (ops.bf16_storage() and x_local.dim() == 2 and x_local.size(1) % 256 == 0 and len(enc.graph_layers) >= 2
                and all(ops.views_ok(n_dec, l.in_channels, l.out_channels, 2) for l in enc.graph_layers[1:])
                and ops.views_ok(n_dec, 256, layer0.out_channels, 2)):
            # both views side by side through every kernel (model/gcl.py GGD._forward_views), exchange per layer
            if ops.features_bf16_ok(x_local):
                x_local = ops.cached_bf16(x_local)
            xw = ops.xw_table(x_local, layer0.lin.weight)                   # [n_local, h] bf16
            if part.comm.live:
                h = _PairedTable.apply(xw, perm_slots, part.comm, ops.pair_rows)    # slot j: [xw_j | xw_perm(j)], whole table
            else:                                   # one rank: nothing to gather -- the single-GPU pairing and its one-launch backward
                h = ops.pair_rows(xw, perm)
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
            if ops.ggd_bce_ok(s[:, 0], s[:, 1]):          # mean over the 2 n_loc local logits, one launch each way
                return ops.ggd_bce(s[:, 0], s[:, 1]) * (2.0 * n_loc * part.world / total)
            pred = torch.cat([s[:, 0], s[:, 1]])
            gt = torch.cat([torch.ones_like(s[:, 0]), torch.zeros_like(s[:, 1])])
            return torch.nn.functional.binary_cross_entropy_with_logits(pred, gt, reduction="sum") * (part.world / total)
    # composable form: the shared first product paired over the whole table, then the two views one after the other
    xw = backend.linear(x_local, layer0.lin.weight)
    hdim = xw.size(1)
    paired = _PairedTable.apply(xw, perm_slots, part.comm,
                                lambda table, ps: torch.cat([table, table[ps]], dim=1))      # [n_table, 2h]
    outs = []
    for v in range(2):
        t = _TableView.apply(paired, v, hdim)                                # this view's table, [n_table, h]
        drop = (dropout_of(n_loc, layer0.out_channels, t.device)
                if (dropout_of is not None and enc.drop_out and enc.training and last > 0) else None)
        h = backend.table_conv(t, layer0.bias, pg, gw, relu=(last > 0), drop=drop)
        for i, layer in enumerate(enc.graph_layers[1:], start=1):
            drop = (dropout_of(n_loc, layer.out_channels, h.device)
                    if (i != last and dropout_of is not None and enc.drop_out and enc.training) else None)
            h = backend.gcn_conv(h, layer.lin.weight, layer.bias, pg, gw, relu=(i != last), drop=drop)
        outs.append(h)
    return _ggd_head(ggd, outs[0], outs[1], backend, total, part.world)


class _TableView(torch.autograd.Function):
    """Columns ``[v h, (v + 1) h)`` of the paired table; the gradient goes back into its half (own rows only are filled
    by the consumer, the rest stays zero)."""

    @staticmethod
    def forward(ctx, paired, v, h):
        ctx.v, ctx.h, ctx.width = v, h, paired.size(1)
        return paired[:, v * h:(v + 1) * h].contiguous()

    @staticmethod
    def backward(ctx, g):
        out = g.new_zeros(g.size(0), ctx.width)
        out[:, ctx.v * ctx.h:(ctx.v + 1) * ctx.h] = g
        return out, None, None


class TripletShare:
    """A rank's share of a triplet list every rank holds (``index [2, T]`` global node ids, ``types [T]``) under
    both-owner scoring: ``own`` -- the head is one of the rank's nodes: the rank scores it, counts it in the loss and in
    d rel_emb -- and ``foreign`` -- the tail is local, the head is not: scored again here, for d score only, so that the
    gradient of the rank's OWN embedding rows is complete without anything coming back over the wire (every triplet is
    scored by the owner of its head and, when that is another rank, by the owner of its tail; counted once).
    ``n_own`` first; ``entries``: positions in ``cat(head, tail)`` of the incidences with a local node when the caller
    knows them by construction (sampling.BlockNegativeSampler), else found with one mask + host read (static positives)."""

    def __init__(self, part, index, types, n_own=None, entries=None, slots=None, types_own=None, types_foreign=None):
        """``slots``: the triplets' table slots when the producer already wrote them (``BlockNegativeSampler`` with
        ``n_max``: ``index`` is then not needed and may be None); ``types_own / types_foreign``: persistent tensors of the
        two id ranges (their relation order is cached ON the tensor object: a fresh slice per step would re-sort)."""
        if n_own is None:
            h_loc, t_loc = part.is_local(index[0]), part.is_local(index[1])
            own = torch.nonzero(h_loc).squeeze(1)
            foreign = torch.nonzero(t_loc & ~h_loc).squeeze(1)
            # a static set (the positives): each range stored in (relation, tail) order -- the scorer then reads the ids
            # in place (their relation order is the identity), consecutive triplets share or neighbour a tail row, and the
            # tail half of a node's incidence list is a few runs of consecutive records (ops._SortedTriplets on one GPU)
            own, foreign = (sel[torch.argsort(types[sel] * part.num_nodes + index[1][sel], stable=True)] for sel in (own, foreign))
            order = torch.cat([own, foreign])
            index, types, n_own = index[:, order], types[order], int(own.numel())
        self.part, self.n_own = part, int(n_own)
        self.index = index.contiguous() if index is not None else None
        self.types = types.contiguous()
        self.slots = slots if slots is not None else part.slots(self.index).contiguous()
        self.entries = entries
        k = self.n_own
        self.types_own = types_own if types_own is not None else self.types[:k].contiguous()
        self.types_foreign = types_foreign if types_foreign is not None else self.types[k:].contiguous()
        self._owned = None

    def owned(self):
        """The ``ops.OwnedTriplets`` of the fused scorer (built once per share: its incidence lists are cached on it)."""
        if self._owned is None:
            from . import ops
            k, part = self.n_own, self.part
            self._owned = ops.OwnedTriplets(self.slots[:, :k], self.types_own, self.slots[:, k:], self.types_foreign,
                                            part.rank * part.n_max if part.comm.live else 0, part.n_local, self.entries,
                                            both=self.slots)
        return self._owned


class _GatherRowsOwnGrad(torch.autograd.Function):
    """forward: every rank's rows to every rank; backward: the rank's OWN rows of the table gradient, nothing else --
    for consumers that complete the gradient of the rank's rows locally (both-owner scoring)."""

    @staticmethod
    def forward(ctx, shard, comm):
        ctx.comm, ctx.n = comm, shard.size(0)
        return comm.all_gather(shard)

    @staticmethod
    def backward(ctx, dtable):
        comm = ctx.comm
        own = comm.own_rows(dtable) if comm.live else dtable
        return own[: ctx.n], None


def _score(table_h, rel_r, table_t, kind):
    """DistMult / TransE logits (reference decoder.py:45-49 / 31-38) in torch -- the composable form's scorer."""
    if kind in (0, "dismult", "distmult"):
        return (table_h * rel_r * table_t).sum(1)
    hn = table_h / table_h.abs().sum(1, keepdim=True).clamp_min(1e-12)
    tn = table_t / table_t.abs().sum(1, keepdim=True).clamp_min(1e-12)
    return -(hn + rel_r - tn).abs().sum(1)


def kge_bce_partitioned(z_local, rel, pos: TripletShare, neg: TripletShare, kind, part, total, reg, backend):
    """BCE over the rank's counted triplets + its nodes' share of the L2 terms (reference kge_module.py:121-129), with the
    gradients of both-owner scoring: d rel_emb from the counted (head-owned) triplets, dz of the rank's rows from EVERY
    triplet that touches them.  ``total`` = global triplet count / world; the returned scalar's mean over ranks is the
    whole-graph loss and the bucket's gradient average the whole-graph gradient.  HipBackend: ONE fused op
    (``ops.kge_bce_loss_owned``); else the composable torch form."""
    if backend.fused:
        from . import ops
        loss, _, _ = ops.kge_bce_loss_owned(z_local, rel, pos.owned(), neg.owned(), kind, part.comm, total, reg=reg,
                                            reg_rows=part.num_nodes / part.world)
        return loss
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    table = _GatherRowsOwnGrad.apply(z_local, part.comm)                     # autograd: own rows only
    table_d, rel_d = table.detach(), rel.detach()
    value, for_z = 0.0, 0.0
    for share, label in ((pos, 1.0), (neg, 0.0)):
        h, t, r, k = share.slots[0], share.slots[1], share.types, share.n_own
        if h.numel() == 0:
            continue
        y = torch.full((h.numel(),), label, dtype=z_local.dtype, device=z_local.device)
        # every incident triplet, gradient into the rank's own rows only (the gather hands back nothing else)
        for_z = for_z + bce(_score(table[h], rel_d[r], table[t], kind), y, reduction="sum")
        # the counted ones, gradient into the relation table
        value = value + bce(_score(table_d[h[:k]], rel[r[:k]], table_d[t[:k]], kind), y[:k], reduction="sum")
    loss = (value + (for_z - for_z.detach() if torch.is_tensor(for_z) else 0.0)) / total
    if reg:
        loss = loss + reg * (z_local.pow(2).sum() / (part.num_nodes / part.world * z_local.size(1)) + rel.pow(2).mean())
    return loss


def kge_loss_partitioned(model, x_local, prg: PartitionedRelGraph, backend, pos: TripletShare, neg: TripletShare, kind,
                         dropout_of=None, n_pos_global=None, n_neg_global=None):
    """``KGEModule.training_step``'s loss (reference kge_module.py:110-129) with the encoder over the rank's shard and
    the scorer over the triplets that touch the rank's nodes (``TripletShare``: both-owner scoring).  The BCE terms are
    normalised by the GLOBAL triplet count over ``world``, so the bucket's average is the whole-graph mean.  The scorer
    reads the all-gathered embedding table (bf16 on the wire in bf16 mode) through slot indices; its gradient needs no
    exchange: each rank reduces the lists of its own nodes (round 3 reduce-scattered an fp32 ``[N, d]`` table here)."""
    part = prg.part
    z_local = rgcn_encoder_partitioned(model.encoder, x_local, prg, backend, dropout_of)
    total = (n_pos_global + n_neg_global) / part.world
    return kge_bce_partitioned(z_local, model.decoder.rel_emb, pos, neg, kind, part, total, 1e-2, backend)


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
    if world == 1 or not dist.is_initialized():      # (a simulated rank reports its own share)
        return loss_local.detach()
    t = loss_local.detach().clone()
    dist.all_reduce(t)
    return t / world
