"""hipGraph replay of the reference's SAMPLED GCL regime: ``train_gcl.py`` trains on ``NeighborLoader`` batches
(``biomedkg/data_module.py:93-99``: 128 seed nodes, fan-out ``[30] * 3``), i.e. on a single-type graph of PrimeKG's
gene/protein size a subgraph of ~25 k nodes and ~220 k edges per step.  Run eagerly that step is ~2.5 ms of kernels under
~3.6 ms of host work (sampler calls, ~300 launches, autograd bookkeeping): the GPU is idle 30 % of the time
(``bench.py sampled_regime_gcl``; profiles/README.md).  Here the step is replayed as ONE captured graph, with the recipe of
``biomedkg_amd/sampled.py`` (fixed-capacity buffers, sizes in device memory, pads that are arithmetically invisible):

* ``NeighborDeviceSampler`` leaves a batch in one int64 vector ``sizes[3] | n_id[N] | src[cap] | dst[cap]`` with no host
  read (``samplers.NeighborSampler.enqueue``); the two sizes travel to pinned host memory asynchronously and are read ONE
  STEP LATER -- the sampler runs a batch ahead of the trainer, on its own stream.
* The host only picks the **edge bucket** (a geometric ladder of capacities) and GGD's augmentation coin
  (``biomedkg/model/gcl.py:74``: a host-side ``torch.rand(1).item()`` in the reference too); one graph per
  (bucket, augmented?) is captured lazily.
* Inside the step nothing depends on a host-side size.  The node capacity is the graph's node count: pad NODES read the
  zero row appended to the feature table and have no edges, so nothing gathers them; pad EDGES are self loops, which ``gcn_norm`` gives weight 0 (``add_remaining_self_loops`` replaces them by the unit loop every node
  gets anyway; they sit on the LAST node, behind the real edges, whose targets ascend: the target-side CSR of a batch
  needs no sort); GGD's corruption permutes the real rows only (``bkg_randperm_dev``: a keyed bijection of ``[0, n_b)`` with
  ``n_b`` read on the device); the BCE sums over the real rows and divides by a device scalar, so the pad rows' logits
  get zero gradient and with them every pad row of every layer.

Arithmetic: the padded step computes ``GGDModule.training_step`` (``biomedkg/gcl_module.py:52-56,229-234``) on the real
part of the batch -- ``tests/test_hip_sampled_gcl.py`` holds it against the ordinary step on the unpadded batch with the
same draws, and a replayed graph against the same padded step run eagerly (bit-equal).

All three GCL models of ``train_gcl.py:15-38`` train this way (round 4).  **DGI** (``gcl_module.py:140-143``): the readout
averages the real rows (a masked column sum over a device count), the corruption permutes the real rows, the JSD terms of
the pad rows are multiplied by zero.  **GRACE** (``gcl_module.py:186-190``): the projected rows of the pads are zeroed
before the NT-Xent, so a pad COLUMN is the zero vector and adds exactly ``exp(0) = 1`` to every real row's denominator --
subtracted again as ``2 (n_cap - n_b)``, a device scalar -- and contributes nothing to any gradient (``E_ij z_j = 0``);
pad ROWS get ``1 / D = 0`` and their gradient rows are cleared (``_MaskedNtXent``: the symmetric NT-Xent kernels of
csrc/ntxent2.hip over the ``2 n_cap`` rows, the same executed flops as the whole-batch loss).  GRACE draws two feature masks and two edge masks per step (in-kernel
Philox / the device generator: graph-safe) and has no host coin: one graph per edge bucket.
"""
from __future__ import annotations

import random

import torch
import torch.nn.functional as F

from . import _lib as L
from . import ops
from . import rand as R
from .graph import as_captured
from .graphs import make_capturable
from .optim import clip_adam_step
from .samplers import NeighborSampler, SubgraphBatch, rows_of

_SIZES = 3     # (unused), nodes, edges -- the sampler's own counter vector


def edge_buckets(e_max, first=4096, ratio=1.25):
    """Edge capacities ``first * ratio^k`` (multiples of 1024) up to ``e_max``: a batch wastes < 25 % of its edge work."""
    out, b = [], float(first)
    while b < e_max:
        out.append(int(-(-b // 1024)) * 1024)
        b *= ratio
    out.append(int(e_max))
    return tuple(sorted(set(min(v, int(e_max)) for v in out)))


class NeighborDeviceSampler:
    """``samplers.NeighborSampler`` with its outputs in ONE int64 vector, so that a consumer takes a private copy of a
    batch with a single launch; ``draw`` enqueues a batch, ``sizes`` waits (a step later) for its two counts."""

    def __init__(self, edge_index, num_nodes, num_neighbors=(30, 30, 30), seed=0):
        self.inner = NeighborSampler(edge_index, num_nodes, num_neighbors, seed)
        if not self.inner.one_read_ok():
            raise ValueError("graph too large for the one-read neighbor sampler")
        dev = edge_index.device
        st = self.inner._state(dev)
        self.num_nodes = int(num_nodes)
        # every node is expanded once and draws distinct in-edges: a batch holds at most min(E, N * fan-out) edges
        self.edge_cap = max(min(st["cap"], edge_index.size(1)), 1)
        self.buf = torch.zeros(_SIZES + self.num_nodes + 2 * st["cap"], dtype=torch.int64, device=dev)
        o = _SIZES + self.num_nodes
        st["ctr"], st["n_id"] = self.buf[:_SIZES], self.buf[_SIZES:o]
        st["e_src"], st["e_dst"] = self.buf[o:o + st["cap"]], self.buf[o + st["cap"]:o + 2 * st["cap"]]
        self.cap = st["cap"]
        self.host = torch.zeros(_SIZES, dtype=torch.int64).pin_memory()
        self.event = torch.cuda.Event()

    def views(self, buf):
        o = _SIZES + self.num_nodes
        return dict(sizes=buf[:_SIZES], n_id=buf[_SIZES:o], src=buf[o:o + self.cap], dst=buf[o + self.cap:o + 2 * self.cap])

    def draw(self, seeds):
        self.inner.enqueue(seeds.contiguous())
        self.host.copy_(self.buf[:_SIZES], non_blocking=True)
        self.event.record()

    def sizes(self):
        self.event.synchronize()
        return int(self.host[1]), int(self.host[2])


class _StepRand(R.DeviceRand):
    """DeviceRand of a padded GGD step: in-kernel stream ids restart at every step (a captured step freezes them; the
    device replay counter tells two steps apart), the corruption permutes the real rows, the coin is the loop's."""

    def __init__(self, seed, n_dev, epoch):
        super().__init__(seed)
        self.n_dev, self.epoch, self.aug, self.last_perm = n_dev, epoch, False, None

    def begin_step(self, aug):
        self.counter, self.aug = 0, bool(aug)

    def coin(self):
        return 0.0 if self.aug else 1.0          # GGD augments when coin < p (p in (0, 1]): decided by the loop, on the host

    def perm(self, n, device):
        out = torch.empty(n, dtype=torch.int64, device=device)
        self.counter += 1
        L.call("bkg_randperm_dev", L.ptr(self.n_dev), n, (self.seed + 0x9E3779B9 * self.counter) & ((1 << 63) - 1),
               L.ptr(self.epoch), L.ptr(out), L.stream())
        self.last_perm = out
        return out


class _MaskedNtXent(torch.autograd.Function):
    """NT-Xent (reference gcl_module.py:171-173: DualBranchContrast(InfoNCE(tau), L2L, intraview_negs)) over the REAL rows of
    a padded batch, through the SYMMETRIC matrix-core kernels of the whole-batch loss (csrc/ntxent2.hip: the same executed
    flops as the unpadded step).  ``h1 / h2 [n_cap, h]`` arrive with their pad rows zeroed; ``w [n_cap]`` (1.0 real, 0.0
    pad) and the real row count ``n_dev`` (int64 device scalar) say which.  A pad column is the zero vector: it adds
    exactly exp(0) = 1 to every row's D -- taken out again as ``2 (n_cap - n_b)`` -- and nothing to any gradient; pad rows
    get 1 / D = 0 and their gradient rows are cleared.  Every size the kernels see is the capacity."""

    @staticmethod
    def forward(ctx, h1, h2, w, n_dev, tau, bf16):
        h1, h2 = h1.contiguous(), h2.contiguous()
        n, h = h1.shape
        m, dev = 2 * n, h1.device
        hpad = L.lib().bkg_ntxent_hpad(h, bf16)
        if hpad < 0:
            raise ValueError(f"ntxent: embedding width {h} > 256 is not supported")
        zn = torch.empty(m, hpad, dtype=torch.float32, device=dev)
        zn16 = torch.empty(m, hpad, dtype=torch.bfloat16, device=dev) if bf16 else None
        norm = torch.empty(m, dtype=torch.float32, device=dev)
        posdot = torch.empty(n, dtype=torch.float32, device=dev)
        L.call("bkg_ntxent_prepare", L.ptr(h1), h1.stride(0), L.ptr(h2), h2.stride(0), n, h, hpad, tau, L.ptr(zn), L.ptr(zn16),
               L.ptr(norm), L.ptr(posdot), L.stream())
        ws = torch.empty(L.lib().bkg_ntxent_forward_workspace_floats(m, h, bf16), dtype=torch.float32, device=dev)
        den = torch.empty(2, m, dtype=torch.float32, device=dev)
        L.call("bkg_ntxent_forward", L.ptr(zn16 if bf16 else zn), bf16, m, h, tau, L.ptr(posdot), L.ptr(ws), L.ptr(den[0]),
               L.ptr(den[1]), None, None, L.stream())
        n_b = n_dev.to(torch.float32)
        w2 = torch.cat([w, w])
        real = w2 > 0
        d_real = torch.where(real, den[0] - 2.0 * (n - n_b), torch.ones_like(den[0]))       # (pads: log 1 = 0)
        rden = torch.where(real, 1.0 / d_real, torch.zeros_like(d_real)).contiguous()
        terms = torch.log(d_real.double()).sum() - 2.0 * (posdot.double() * w.double()).sum() / tau
        ctx.tau, ctx.n, ctx.h, ctx.bf16 = tau, n, h, bf16
        ctx.save_for_backward(zn, zn16, norm, rden, real, n_b)
        return (terms / (2.0 * n_b.double())).float().reshape(())

    @staticmethod
    def backward(ctx, g):
        zn, zn16, norm, rden, real, n_b = ctx.saved_tensors
        n, h, tau, bf16 = ctx.n, ctx.h, ctx.tau, ctx.bf16
        m, dev = 2 * n, zn.device
        # the kernel's own mean runs over all 2 n_cap rows: rescale to the 2 n_b real ones (a device scalar)
        gout = (g.reshape(1).float() * (float(n) / n_b)).contiguous()
        ws = torch.empty(L.lib().bkg_ntxent_backward_workspace_floats(m, h, bf16), dtype=torch.float32, device=dev)
        dh = torch.empty(2, n, h, dtype=torch.float32, device=dev)
        L.call("bkg_ntxent_backward", L.ptr(zn16 if bf16 else zn), bf16, L.ptr(zn), L.ptr(norm), L.ptr(rden), L.ptr(gout), m, h,
               tau, L.ptr(ws), L.ptr(dh[0]), h, L.ptr(dh[1]), h, L.stream())
        dh = torch.where(real.view(2, n, 1), dh, torch.zeros_like(dh))       # (a pad row's normalisation divides by eps)
        return dh[0], dh[1], None, None, None, None


GRACE_NODE_STEP = int(__import__("os").environ.get("BKG_GRACE_NODE_STEP", "512"))   # rows per rung of GRACE's node ladder


class NeighborGraphedGcl:
    """``GGDModule`` / ``DGIModule`` / ``GRACEModule`` training steps over NeighborLoader batches, one hipGraph replay each
    (see the module docstring).

    ``step()`` = sample (one batch ahead) + zero_grad + forward + loss + backward + clip + Adam; returns the loss as a
    device scalar (the replay's static output; read it before the next step).  ``graphs=False`` runs the same padded step
    eagerly (the bit-equality check of the tests).  One epoch = ``len(self)`` steps over a shuffled seed order."""

    def __init__(self, module, x, edge_index, batch_size=128, num_neighbors=(30, 30, 30), optimizer=None, clip=1.0, seed=0,
                 shuffle=True, buckets=None, graphs=True, max_graphs=16):
        L.require_cuda(x, edge_index)
        from .model.gcl import DGI, GGD, GRACE
        kinds = {GGD: "ggd", DGI: "dgi", GRACE: "grace"}
        self.kind = kinds.get(type(getattr(module, "model", None)))
        if self.kind is None:
            raise ValueError("the padded NeighborLoader step is built for the GGD, DGI and GRACE modules")
        self.module, self.clip, self.graphs_on = module, clip, graphs
        self.max_graphs = int(max_graphs)     # every captured graph keeps its step's intermediates in a pool of its own
        dev = x.device
        self.sampler = NeighborDeviceSampler(edge_index, x.size(0), num_neighbors, seed)
        self.n_cap, self.e_cap = self.sampler.num_nodes, self.sampler.edge_cap
        # (a ladder handed in is cut at the most edges a batch can hold: the sampler's buffers end there)
        self.buckets = tuple(sorted({min(int(b), self.e_cap) for b in (buckets or edge_buckets(self.e_cap))}))
        if self.buckets[-1] < self.e_cap:
            self.buckets += (self.e_cap,)
        self.batch_size, self.shuffle = int(batch_size), shuffle
        self.x = x
        self.x_ext = torch.cat([x, x.new_zeros((1,) + tuple(x.shape[1:]))])           # + the zero row pad nodes read
        self.cur = torch.zeros_like(self.sampler.buf)                                   # the batch the step reads
        self.v = self.sampler.views(self.cur)
        self.iota_n = torch.arange(self.n_cap, device=dev)
        self.iota_e = torch.arange(self.buckets[-1], device=dev)
        self.last_node = torch.full((), self.n_cap - 1, dtype=torch.int64, device=dev)   # pad edges: its self loops
        self.pad_row = torch.full((), self.n_cap, dtype=torch.int64, device=dev)
        if optimizer is None:         # the optimizer of BaseGCL.configure_optimizers, without the trainer-bound scheduler
            optimizer = torch.optim.Adam(list(module.model.parameters()), lr=module.lr, fused=True)
        self.optimizer = make_capturable(optimizer)
        self.epoch = torch.zeros(1, dtype=torch.int32, device=dev)
        self.rand = _StepRand(seed, self.v["sizes"][1:2], self.epoch)
        self.coins = random.Random(int(seed) * 7919 + 1)
        self.execs, self.losses = {}, {}
        self.stats = {"replayed": 0, "captured": 0, "eager_padded": 0}
        self.last_sizes = self.last_aug = None
        self._order, self._pos, self._primed = None, 0, False
        self._gen = torch.Generator(device=dev).manual_seed(int(seed) + 17)
        # everything runs on ONE side stream (autograd's AccumulateGrad nodes must not be born on the default stream if a
        # later capture is to meet them); the sampler on its own, a batch ahead
        self.stream, self.sampler_stream = torch.cuda.Stream(), torch.cuda.Stream()
        self.copied = torch.cuda.Event()
        self._pinned = []

    def __len__(self):
        return (self.n_cap + self.batch_size - 1) // self.batch_size

    def close(self):
        for st in (self.sampler_stream, self.stream):
            st.synchronize()

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass

    # ---- seeds: DataLoader(shuffle=True) over the node ids, batches of ``batch_size``, the tail batch shorter
    def _next_seeds(self):
        if self._order is None or self._pos >= self.n_cap:
            dev = self.x.device
            self._order = (torch.randperm(self.n_cap, device=dev, generator=self._gen) if self.shuffle
                           else torch.arange(self.n_cap, device=dev))
            self._pos = 0
        seeds = self._order[self._pos:self._pos + self.batch_size]
        self._pos += self.batch_size
        return seeds

    # ---- the padded step (captured once per edge bucket and augmentation branch)
    def padded_batch(self, e_b):
        v = self.v
        dev = self.x.device
        # pads: node ids past the count read the zero row; edges past the count are self loops of the LAST node (weight 0
        # under gcn_norm for any node, real or pad) -- behind the real edges, whose targets ascend (the sampler expands
        # frontier nodes in local-id order), so the whole list stays sorted by target and the target-side CSR needs no
        # sort (graph.Graph, _bkg_dst_sorted).  One launch (it was six torch expressions in every captured step).
        n_id = torch.empty(self.n_cap, dtype=torch.int64, device=dev)
        ei = torch.empty(2, e_b, dtype=torch.int64, device=dev)            # fresh tensor: no structure cached on it survives the step
        real_n = torch.empty(self.n_cap, dtype=torch.bool, device=dev)
        L.call("bkg_padded_gcl_batch", L.ptr(v["sizes"]), L.ptr(v["n_id"]), L.ptr(v["src"]), L.ptr(v["dst"]), self.n_cap, e_b,
               self.n_cap, self.n_cap - 1, L.ptr(n_id), L.ptr(ei[0]), L.ptr(ei[1]), L.ptr(real_n), L.stream())
        ei._bkg_dst_sorted = ei._version
        return n_id, ei, real_n

    def padded_loss(self, e_b, n_pad=None):
        """The module's ``training_step`` over the batch in ``self.cur`` padded to ``n_cap`` nodes / ``e_b`` edges: GGD --
        fusion -> GGD -> BCE with logits, mean over the 2 n_b real logits; DGI -- masked readout, JSD over the real rows;
        GRACE -- two augmented views, projector, NT-Xent over the real rows."""
        m = self.module
        n_id, ei, real_n = self.padded_batch(e_b)
        x = m.fusion_fn(x=rows_of(self.x_ext, n_id))
        if self.kind == "dgi":
            return self._dgi_loss(m, x, ei, real_n)
        if self.kind == "grace":
            return self._grace_loss(m, x, ei, real_n, n_pad)
        pos_h, neg_h = m.model(x, ei)
        if ops.ggd_bce_ok(pos_h, neg_h):          # rows below sizes[1] count, the mean divides by 2 sizes[1] (csrc/gcl_loss.hip)
            return ops.ggd_bce(pos_h, neg_h, self.v["sizes"][1:2])
        w = real_n.to(pos_h.dtype)
        inv = 0.5 / self.v["sizes"][1].to(pos_h.dtype)
        return (F.binary_cross_entropy_with_logits(pos_h, torch.ones_like(pos_h), weight=w, reduction="sum")
                + F.binary_cross_entropy_with_logits(neg_h, torch.zeros_like(neg_h), weight=w, reduction="sum")) * inv

    def _dgi_loss(self, m, x, ei, real_n):
        """``DGIModule.calculate_loss`` (reference gcl_module.py:140-143, gcl.py:23-27) on the real rows."""
        import math
        w = real_n.to(torch.float32)
        n_b = self.v["sizes"][1].to(torch.float32)
        z, g, zn = m.model(x, ei, summary=lambda t: torch.sigmoid((t.float() * w[:, None]).sum(0, keepdim=True) / n_b))
        sim_pos = ops.linear(z, g).squeeze(1)
        sim_neg = ops.linear(zn, g).squeeze(1)
        log2 = math.log(2.0)
        e_pos = ((log2 - F.softplus(-sim_pos)) * w).sum() / n_b
        e_neg = ((F.softplus(-sim_neg) + sim_neg - log2) * w).sum() / n_b
        return e_neg - e_pos

    def _grace_loss(self, m, x, ei, real_n, n_pad=None):
        """``GRACEModule.calculate_loss`` (reference gcl_module.py:186-190) on the real rows.  The loss is quadratic in the
        rows it is given: the encoder runs over the ``n_cap`` padded nodes, NT-Xent over the first ``n_pad`` of them only
        (the real rows come first; ``n_pad`` >= the batch's node count, a rung of ``GRACE_NODE_STEP`` -- the host knows
        the count: it picks the edge bucket from it): at most ~4 % more flops than the eager loop's exact-size loss,
        instead of the (n_cap / n_b)^2 - 1 ~ 15 % that made a replayed fp32 step slower than an eager one."""
        w = real_n.to(torch.float32)
        z1, z2 = m.model.views(x, ei)
        h1, h2 = [m.model.project(z).float() * w[:, None] for z in (z1, z2)]
        if n_pad is not None and n_pad < h1.size(0):
            h1, h2, w = h1[:n_pad], h2[:n_pad], w[:n_pad]
        return _MaskedNtXent.apply(h1, h2, w, self.v["sizes"][1:2], float(m.contrast_model.tau),
                                   int(ops.get_matmul_precision() == "bf16"))

    def _padded_step(self, e_b, aug, n_pad=None):
        with as_captured():               # eager or recorded: the same kernels over the same (untightened) tables
            self.epoch.add_(1)
            self.rand.begin_step(aug)
            self.optimizer.zero_grad(set_to_none=True)
            loss = self.padded_loss(e_b, n_pad)
            loss.backward()
            clip_adam_step(self.optimizer, self.clip)
        return loss

    def unpadded_batch(self):
        """The batch in ``self.cur`` as the eager loader would hand it out (tests)."""
        n_b, e_b = self.last_sizes
        v = self.v
        n_id = v["n_id"][:n_b].clone()
        return SubgraphBatch(rows_of(self.x_ext, n_id), torch.stack([v["src"][:e_b], v["dst"][:e_b]]), None,
                             min(self.batch_size, n_b), n_id, None)

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
            self.close()                  # drain the loop's streams before anybody drops its buffers
            raise
        finally:
            L.call("bkg_set_rng_epoch", None)
            if trainer is not None:
                self.module.__dict__["trainer"] = trainer

    def _draw_ahead(self, main):
        with torch.cuda.stream(self.sampler_stream):
            self.sampler_stream.wait_event(self.copied)
            self.sampler.draw(self._next_seeds())

    def _step(self):
        s, main = self.sampler, torch.cuda.current_stream()
        if not self._primed:
            self.copied.record(main)
            self._draw_ahead(main)
            self._primed = True
        n_b, e_b = s.sizes()                                               # of the batch drawn one step ago
        self.last_sizes = (n_b, e_b)
        main.wait_event(s.event)
        self.cur.copy_(s.buf)                                              # private copy, then the sampler moves on:
        self.copied.record(main)                                           # the next batch is drawn UNDER this step
        self._draw_ahead(main)
        bucket = next(b for b in self.buckets if b >= max(e_b, 1))
        # GGD's augmentation coin (host side in the reference too); DGI and GRACE have none: one graph per bucket
        aug = self.last_aug = (self.coins.random() < self.module.model.p) if self.kind == "ggd" else False
        # GRACE: the rung of the node ladder its NT-Xent runs over (None: every row -- the other losses are linear in the rows)
        n_pad = (min(self.n_cap, -(-max(int(n_b), 1) // GRACE_NODE_STEP) * GRACE_NODE_STEP) if self.kind == "grace" else None)
        if not self.graphs_on:
            self.stats["eager_padded"] += 1
            return self._padded_step(bucket, aug, n_pad).detach()
        key = (bucket, aug, n_pad)
        g = self.execs.get(key)
        if g is None and len(self.execs) >= self.max_graphs:      # a regime that wanders over many buckets: the rest stay eager
            self.stats["eager_padded"] += 1
            return self._padded_step(bucket, aug, n_pad).detach()
        if g is None:
            # first batch of this kind: one eager padded step (it IS this batch's training step; it also warms the
            # allocator and the kernel plans), then the capture pass, which records and does not execute
            loss = self._padded_step(bucket, aug, n_pad).detach().clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            # a captured graph reads the fuser's cached table output by address: keep those tensors alive with the loop
            for hit in getattr(self.module.modality_transform, "__dict__", {}).get("_fused_cache", []):
                if all(hit[1] is not t for t in self._pinned):
                    self._pinned.append(hit[1])
            try:
                with torch.cuda.graph(g):
                    self.losses[key] = self._padded_step(bucket, aug, n_pad).detach()
            except RuntimeError as why:      # a module on the path that cannot be captured: same steps, run eagerly
                torch.cuda.synchronize()
                self.graphs_on = False
                self.optimizer.zero_grad(set_to_none=True)
                print(f"[sampled-gcl] padded steps stay eager: {str(why).splitlines()[0]}", file=__import__("sys").stderr, flush=True)
                return loss
            self.execs[key] = g
            self.stats["captured"] += 1
            return loss
        g.replay()
        self.stats["replayed"] += 1
        ops.note_graph_replay()          # weights moved under an unchanged tensor version (ops._weight_bf16)
        return self.losses[key]
