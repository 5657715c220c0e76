"""hipGraph replay of the reference's SAMPLED KGE regime: ``train_kge.py`` trains on GraphSAINT random-walk batches
(``biomedkg/data_module.py:100-107``: ``batch_size`` walks of length 10, 1000 steps per epoch), i.e. subgraphs of at
most ``batch_size * 11`` nodes and a few thousand edges -- ~270 launches of a few microseconds each per step, so the
step time is launch latency and three host round trips (the sampler's ``unique`` / ``nonzero``, the segment count of
the relational structure, the negative sampler's survivor count), not arithmetic.

A captured graph freezes kernel arguments and grids, while every size of such a batch is data dependent.  The way out
here: **fixed-capacity buffers, sizes in device memory, pads that are arithmetically invisible**.

* ``SaintDeviceSampler`` draws a batch with no host round trip (``bkg_saint_batch``): node ids padded to
  ``batch_size * (walk_length + 1) + 1`` with the index of a ZERO row appended to the feature table, edges padded to a
  capacity with self loops of the batch's last local node (the *dummy* node, always a pad), the three sizes (nodes,
  edges, loop-free edges) in a device vector that is also copied to pinned host memory -- asynchronously; the host
  reads it one step later, when it has long arrived (the sampler runs one batch ahead of the trainer, on its own stream).
* The host only picks the **edge bucket** (a geometric ladder of capacities); one graph per bucket is captured lazily.
* Inside the step nothing depends on a host-side size: pad nodes are isolated zero-feature rows; pad edges only feed the
  dummy node, which no triplet touches (their sources and types are spread, so no row of any structure grows long); the relational structure always carries ``E_bucket`` segments (the surplus
  ones empty); the negative sampler reads its extents from the device vector (``bkg_negative_sample_dev``); pad triplets
  carry a relation id one past the real ones, which the fused scorer / loss pass never visits, and zero gradient
  records; the BCE mean divides by a device scalar; the L2 term masks the pad rows.
* Batches that fall outside the captured regime (more edges than the largest bucket, or so dense that PyG's sampler
  would need its later rounds) take the ordinary eager step on the unpadded batch.

Arithmetic: the padded step computes the reference's ``KGEModule.training_step`` (``biomedkg/kge_module.py:104-131``)
on the real part of the batch -- ``tests/test_hip_sampled.py`` holds it against the unpadded eager step on the same
batch and negatives, and a replayed graph against the same padded step run eagerly (bit-equal).
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops
from . import rand as R
from .graph import RelGraph, as_captured, graph_of, plant_rel_graph
from .graphs import make_capturable
from .optim import clip_adam_step
from .samplers import SubgraphBatch, _relabel, rows_of

EDGE_BUCKETS = (1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 49152, 65536)
_SIZES = 3     # nodes, edges, loop-free edges


def choose_bucket(buckets, neg_ratio, n_b, e_b, e_free):
    """The edge bucket a batch of ``n_b`` nodes / ``e_b`` edges (``e_free`` of them with head != tail) replays under, or
    None when it must take the ordinary eager step: more edges than the largest bucket, or a batch on which PyG's
    ``negative_sampling`` would leave its first round (``kge_module.py:93``: all cells enumerated because
    ``int(1.1 k / prob)`` reaches the population, or so dense -- ``prob <= 0.9`` -- that later rounds could be needed)."""
    if n_b < 2 or e_b < 1 or e_b > buckets[-1]:
        return None
    population = n_b * (n_b - 1)
    prob = 1.0 - e_free / population
    k = neg_ratio * e_b
    if prob <= 0.9 or int(1.1 * k / prob) >= population:
        return None
    # PyG's first round keeps ~1.1 k of its int(1.1 k / prob) candidates (binomial, sigma^2 <= 1.1 k (1 - prob)); the
    # captured step has only that round, so the 10 % margin must be worth six sigma -- tiny dense batches go eager
    if 0.1 * k < 6.0 * (1.1 * k * (1.0 - prob)) ** 0.5:
        return None
    return next(b for b in buckets if b >= e_b)


class SaintDeviceSampler:
    """GraphSAINT random-walk batches into fixed-capacity device buffers (see the module docstring).

    ``buf`` is ONE int64 vector holding ``sizes[3] | n_id[node_cap] | src[edge_cap] | dst | type | eid`` so that a
    consumer can take a private copy with a single launch."""

    def __init__(self, edge_index, edge_type, num_nodes, batch_size, walk_length=10, seed=0, edge_cap=EDGE_BUCKETS[-1],
                 pad_types=1):
        L.require_cuda(edge_index, edge_type)
        self.pad_types = int(pad_types)
        dev = edge_index.device
        self.edge_index, self.edge_type = edge_index.contiguous(), edge_type.contiguous()
        self.graph = graph_of(self.edge_index, num_nodes)
        self.num_nodes, self.num_edges = int(num_nodes), self.edge_index.size(1)
        self.batch_size, self.walk_length, self.seed = int(batch_size), int(walk_length), int(seed)
        self.node_cap = self.batch_size * (self.walk_length + 1) + 1      # + the dummy node (always a pad)
        self.edge_cap = int(edge_cap)
        self.pad_node = self.node_cap - 1
        self.buf = torch.zeros(_SIZES + self.node_cap + 4 * self.edge_cap, dtype=torch.int64, device=dev)
        self.host = torch.zeros(_SIZES, dtype=torch.int64).pin_memory()
        self.start = torch.zeros(self.batch_size, dtype=torch.int64, device=dev)
        self.walks = torch.zeros(self.batch_size, self.walk_length + 1, dtype=torch.int64, device=dev)
        i32 = lambda k: torch.zeros(k, dtype=torch.int32, device=dev)     # noqa: E731
        self._node = [i32(self.num_nodes) for _ in range(3)]
        self._edge = [i32(self.num_edges) for _ in range(2)]
        self._ws = torch.empty(L.lib().bkg_saint_batch_workspace_bytes(self.num_nodes, self.num_edges), dtype=torch.uint8,
                               device=dev)
        self.epoch = torch.zeros(1, dtype=torch.int32, device=dev)        # replay counter mixed into the walk seed
        self.event = torch.cuda.Event()
        self.graph_exec = None

    @staticmethod
    def views(buf, node_cap, edge_cap):
        o = _SIZES + node_cap
        return dict(sizes=buf[:_SIZES], n_id=buf[_SIZES:o], src=buf[o:o + edge_cap], dst=buf[o + edge_cap:o + 2 * edge_cap],
                    type=buf[o + 2 * edge_cap:o + 3 * edge_cap], eid=buf[o + 3 * edge_cap:o + 4 * edge_cap])

    def _draw(self, start=None, use_epoch=True):
        v = self.views(self.buf, self.node_cap, self.edge_cap)
        if start is None:
            self.epoch.add_(1)                                # roots come from the kernel's own keyed stream
        else:
            self.start.copy_(start)
        csr = self.graph.by_src
        L.call("bkg_saint_batch", L.ptr(csr.rowptr), L.ptr(csr.col), L.ptr(self.start) if start is not None else None,
               self.batch_size, self.walk_length,
               self.seed, L.ptr(self.epoch) if use_epoch else None, None, L.ptr(self.walks), L.ptr(self.edge_index[0]),
               L.ptr(self.edge_index[1]), L.ptr(self.edge_type), self.num_nodes, self.num_edges, L.ptr(self._node[0]),
               L.ptr(self._node[1]), L.ptr(self._node[2]), L.ptr(self._edge[0]), L.ptr(self._edge[1]), self.node_cap,
               self.edge_cap, self.pad_node, self.pad_types, L.ptr(v["n_id"]), L.ptr(v["src"]), L.ptr(v["dst"]), L.ptr(v["type"]),
               L.ptr(v["eid"]), L.ptr(v["sizes"]), L.ptr(self._ws), self._ws.numel(), L.stream())

    def draw(self, start=None, use_epoch=True, graph=True):
        """Enqueue one batch on the current stream (no host wait); ``sizes()`` later waits for its three numbers.
        ``start`` (test hook): the walks' roots instead of fresh uniform ones.  ``graph``: the eleven launches replayed
        as one captured graph (the copy of the sizes to pinned memory stays outside it)."""
        if start is not None or not graph:
            self._draw(start, use_epoch)
        else:
            if self.graph_exec is None:
                self._draw()                                  # eager once (plans, allocator), then captured
                torch.cuda.synchronize()
                self.graph_exec = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_exec):
                    self._draw()
                self.epoch.zero_()                            # the warm-up draw does not count: same stream as eager
            self.graph_exec.replay()
        self.host.copy_(self.buf[:_SIZES], non_blocking=True)
        self.event.record()

    def sizes(self):
        self.event.synchronize()
        return tuple(int(v) for v in self.host)


class _StepRand(R.DeviceRand):
    """DeviceRand whose in-kernel stream ids restart at every step: a captured step freezes them, the device replay
    counter (``bkg_set_rng_epoch``) is what tells two steps apart -- eagerly and replayed alike."""

    def begin_step(self):
        self.counter = 0

    def coin(self):
        raise RuntimeError("a padded KGE step draws no host-side coin")


class SaintGraphedKge:
    """``KGEModule`` training steps over GraphSAINT batches, one hipGraph replay each (see the module docstring).

    ``step()`` = sample (one batch ahead) + zero_grad + forward + loss + backward + clip + Adam; returns the loss as a
    device scalar (the replay's static output; read it before the next step).  ``graphs=False`` runs the same padded
    step eagerly (the bit-equality check of the tests)."""

    def __init__(self, module, x, edge_index, edge_type, batch_size, walk_length=10, optimizer=None, clip=1.0, seed=0,
                 buckets=EDGE_BUCKETS, graphs=True):
        L.require_cuda(x, edge_index, edge_type)
        if getattr(module.model.decoder, "kind", None) is None:
            raise ValueError("the padded step is built on the fused DistMult / TransE loss")
        self.module, self.clip, self.graphs_on = module, clip, graphs
        self.neg_ratio = int(module.neg_ratio) if module.neg_ratio else 1
        dev = x.device
        self.buckets = tuple(sorted(int(b) for b in buckets))
        self.num_relations = module.model.decoder.rel_emb.size(0)
        self.sampler = SaintDeviceSampler(edge_index, edge_type, x.size(0), batch_size, walk_length, seed,
                                          edge_cap=self.buckets[-1], pad_types=self.num_relations)
        self.x = x
        self.x_ext = torch.cat([x, x.new_zeros((1,) + tuple(x.shape[1:]))])           # + the zero row pads point at
        self.n_cap, self.e_cap = self.sampler.node_cap, self.sampler.edge_cap
        self.cur = torch.zeros_like(self.sampler.buf)                                   # the batch the step reads
        self.v = SaintDeviceSampler.views(self.cur, self.n_cap, self.e_cap)
        self.iota_e = torch.arange(self.e_cap, device=dev)
        if optimizer is None:         # the optimizer of KGEModule.configure_optimizers, without the trainer-bound scheduler
            optimizer = torch.optim.Adam(list(module.model.parameters()), lr=module.lr, fused=True)
        self.optimizer = make_capturable(optimizer)
        self.params = [p for g in self.optimizer.param_groups for p in g["params"]]
        self.rand = _StepRand(seed)
        self.epoch = torch.zeros(1, dtype=torch.int32, device=dev)
        self.seed = int(seed) & ((1 << 62) - 1)
        self.execs, self.losses = {}, {}
        self.stats = {"replayed": 0, "captured": 0, "eager_padded": 0, "eager_fallback": 0}
        self.last_sizes = None
        self._primed = False
        # everything -- sampler, eager steps, captures, replays -- runs on ONE side stream: autograd's AccumulateGrad
        # nodes must not be born on the default stream if a later capture is to meet them (torch.cuda.graphs notes)
        self.stream = torch.cuda.Stream()
        self.side, self.side2, self.sampler_stream = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        self.copied = torch.cuda.Event()
        self.branches = True
        self._pinned = []

    def close(self):
        """Wait for everything this loop has in flight (the sampler runs a batch ahead on its own stream): its buffers
        must not go back to the allocator under a running kernel."""
        for st in (self.sampler_stream, self.side, self.side2, self.stream):
            st.synchronize()

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass

    # ---- the padded step (captured once per edge bucket)
    def _negatives(self, ei, et, e_b):
        dev, k_cap = ei.device, self.neg_ratio * e_b
        cand = L.lib().bkg_negative_sample_dev_candidates(k_cap)
        cap = 1 << max(10, (2 * e_b - 1).bit_length())
        table = torch.empty(cap, dtype=torch.int64, device=dev)
        keys = torch.empty(cand, dtype=torch.int64, device=dev)
        flag = torch.empty(2, cand, dtype=torch.int32, device=dev)
        ws = torch.empty(L.lib().bkg_negative_sample_workspace_bytes(cand), dtype=torch.uint8, device=dev)
        neg = torch.empty(2, k_cap, dtype=torch.int64, device=dev)
        neg_type = torch.empty(k_cap, dtype=torch.int64, device=dev)
        count = torch.empty(1, dtype=torch.int64, device=dev)
        L.call("bkg_negative_sample_dev", L.ptr(self.v["sizes"]), self.neg_ratio, L.ptr(ei[0]), L.ptr(ei[1]), L.ptr(et), e_b,
               self.seed, L.ptr(self.epoch), L.ptr(table), cap, L.ptr(keys), L.ptr(flag[0]), L.ptr(flag[1]), k_cap,
               self.num_relations, L.ptr(neg[0]), L.ptr(neg[1]), L.ptr(neg_type) if self.module.neg_ratio else None,
               L.ptr(count), L.ptr(ws), ws.numel(), L.stream())
        self.last_negatives = (neg, neg_type, count)
        return neg, neg_type, count

    def padded_loss(self, e_b):
        """The loss of ``KGEModule.training_step`` over the batch in ``self.cur`` padded to ``e_b`` edges.

        Two branches that only meet at the loss: the relational structure + encoder on the current stream; the
        negatives and every triplet-side structure (relation orders, incidence lists) on a side stream -- in a captured
        graph they become parallel branches, and a step of ~250 tiny launches is bound by the longest chain."""
        m, v = self.module, self.v
        r = self.num_relations
        ei = torch.stack([v["src"][:e_b], v["dst"][:e_b]])                 # fresh tensors: nothing cached survives a step
        et = v["type"][:e_b].clone()
        if m._fix_edge_id is not None:
            et = torch.full_like(et, m._fix_edge_id)
        main = torch.cuda.current_stream()
        side, side2 = (self.side, self.side2) if self.branches else (main, main)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            neg, neg_type, count = self._negatives(ei, et, e_b)
            if not m.neg_ratio:                                            # reference: the positives' own relation ids
                neg_type = torch.where(self.iota_e[:e_b] < count, et, r)
            ei_loss = torch.empty_like(ei)
            et_loss = torch.empty_like(et)
            inv_total = torch.empty(1, dtype=torch.float32, device=ei.device)
            L.call("bkg_padded_loss_inputs", L.ptr(v["sizes"]), L.ptr(count), L.ptr(ei[0]), L.ptr(ei[1]), L.ptr(et), e_b,
                   self.n_cap, r, L.ptr(ei_loss[0]), L.ptr(ei_loss[1]), L.ptr(et_loss), L.ptr(inv_total), L.stream())
            ops.kge_prepare_structures(ei_loss, et_loss, neg, neg_type, self.n_cap, r + 1)
        x = m.fusion_fn(x=rows_of(self.x_ext, v["n_id"]))
        rg = plant_rel_graph(RelGraph(ei, et, self.n_cap, r, static_segments=True, defer_source_side=True))
        side2.wait_stream(main)
        with torch.cuda.stream(side2):                                     # only the backward reads it: third branch
            rg.build_source_side()
        z = m.model.encode(x, ei, et)
        main.wait_stream(side)
        main.wait_stream(side2)
        dec = m.model.decoder
        loss, pos_pred, neg_pred = ops.kge_bce_loss(z, dec.rel_emb, ei_loss, et_loss, neg, neg_type, dec.kind, reg=1e-2,
                                                    inv_total=inv_total, rows=v["sizes"][:1])
        return loss                                                        # BCE + 1e-2 L2, reference kge_module.py:127-129

    def _padded_step(self, e_b):
        with as_captured():               # eager or recorded: the same kernels over the same (untightened) tables
            self.epoch.add_(1)
            self.rand.begin_step()
            self.optimizer.zero_grad(set_to_none=True)
            loss = self.padded_loss(e_b)
            loss.backward()
            clip_adam_step(self.optimizer, self.clip)
        return loss

    # ---- the ordinary step on the unpadded batch (outside the captured regime)
    def unpadded_batch(self, n_b, e_b):
        v = self.v
        n_id = v["n_id"][:n_b].clone()
        if e_b > self.e_cap:
            # the sampler stores the first ``edge_cap`` induced edges only (and reports the true total): the node set is
            # complete, so the induced subgraph is rebuilt from it -- the eager sampler's own torch-glue path
            # (samplers.SaintRandomWalkSampler.sample_with_torch_glue), same edge order (ascending edge id)
            s = self.sampler
            mask = torch.zeros(s.num_nodes, dtype=torch.bool, device=n_id.device)
            mask[n_id] = True
            e_id = torch.nonzero(mask[s.edge_index[0]] & mask[s.edge_index[1]]).squeeze(1)
            if e_id.numel() != e_b:
                raise RuntimeError(f"induced-edge count {e_id.numel()} != the sampler's {e_b}")
            src, dst = _relabel(n_id, s.num_nodes, s.edge_index[0][e_id], s.edge_index[1][e_id])
            return SubgraphBatch(rows_of(self.x_ext, n_id), torch.stack([src, dst]), s.edge_type[e_id].clone(), n_b, n_id, e_id)
        ei = torch.stack([v["src"][:e_b], v["dst"][:e_b]])
        # (rows of the SAME table object as the padded steps: one fused copy of it serves both)
        return SubgraphBatch(rows_of(self.x_ext, n_id), ei, v["type"][:e_b].clone(), n_b, n_id, v["eid"][:e_b].clone())

    def _eager_step(self, n_b, e_b):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.module.training_step(self.unpadded_batch(n_b, e_b))
        loss.backward()
        clip_adam_step(self.optimizer, self.clip)
        return loss.detach()

    def bucket_of(self, n_b, e_b, e_free):
        return choose_bucket(self.buckets, self.neg_ratio, n_b, e_b, e_free)

    def step(self):
        trainer = self.module.__dict__.pop("trainer", None)                # .log() inside a step would sync
        L.call("bkg_set_rng_epoch", L.ptr(self.epoch))
        outer = torch.cuda.current_stream()
        self.stream.wait_stream(outer)
        try:
            with R.use(self.rand), torch.cuda.stream(self.stream):
                loss = self._step()
            outer.wait_stream(self.stream)                                 # the caller reads the loss on its own stream
            return loss
        except BaseException:
            # a failed step (e.g. the reference's own shape error on a degenerate batch) leaves work in flight on the
            # loop's streams: drain them, so that whoever drops this object does not free buffers under running kernels
            self.close()
            raise
        finally:
            L.call("bkg_set_rng_epoch", None)
            if trainer is not None:
                self.module.__dict__["trainer"] = trainer

    def _step(self):
        s, main = self.sampler, torch.cuda.current_stream()
        if not self._primed:
            self.sampler_stream.wait_stream(main)
            with torch.cuda.stream(self.sampler_stream):
                s.draw(graph=self.graphs_on)
            self._primed = True
        n_b, e_b, e_free = s.sizes()                                       # of the batch drawn one step ago
        self.last_sizes = (n_b, e_b, e_free)
        main.wait_event(s.event)
        self.cur.copy_(s.buf)                                              # private copy, then the sampler moves on:
        self.copied.record(main)                                           # the next batch is drawn UNDER this step
        with torch.cuda.stream(self.sampler_stream):
            self.sampler_stream.wait_event(self.copied)
            s.draw(graph=self.graphs_on)
        bucket = self.bucket_of(n_b, e_b, e_free)
        if bucket is None:
            self.stats["eager_fallback"] += 1
            return self._eager_step(n_b, e_b)
        if not self.graphs_on:
            self.stats["eager_padded"] += 1
            return self._padded_step(bucket).detach()
        g = self.execs.get(bucket)
        if g is None:
            # first batch of this bucket: one eager padded step (it IS this batch's training step; it also warms the
            # allocator and the kernel plans), then the capture pass, which records and does not execute
            loss = self._padded_step(bucket).detach().clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            # a captured graph reads the fuser's cached table output by address: keep those tensors alive with the loop
            for hit in getattr(self.module.modality_transform, "__dict__", {}).get("_fused_cache", []):
                if all(hit[1] is not t for t in self._pinned):
                    self._pinned.append(hit[1])
            try:
                with torch.cuda.graph(g):
                    self.losses[bucket] = self._padded_step(bucket).detach()
            except RuntimeError as why:      # a module on the path that cannot be captured: same steps, run eagerly
                torch.cuda.synchronize()
                self.graphs_on = False
                self.optimizer.zero_grad(set_to_none=True)
                print(f"[sampled] padded steps stay eager: {str(why).splitlines()[0]}", file=__import__("sys").stderr, flush=True)
                return loss
            self.execs[bucket] = g
            self.stats["captured"] += 1
            return loss
        g.replay()
        self.stats["replayed"] += 1
        ops.note_graph_replay()          # weights moved under an unchanged tensor version (ops._weight_bf16)
        return self.losses[bucket]
