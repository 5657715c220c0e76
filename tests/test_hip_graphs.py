"""hipGraph capture of a whole training step (biomedkg_amd/graphs.py) for the launch-bound small-graph regime
(BASELINE configs[1]: GRACE on the 5k / 50k graph): a replayed step must be the eager step -- same kernels, same order,
bit for bit when the randomness is pinned --, replays must draw fresh masks (torch's graph-safe generator for the
feature / edge masks, the device replay epoch for the library's in-kernel Philox dropout), and steps that cannot be
captured (host-side coin, sort-based permutation, host-counted negative sampler) must be refused loudly.
Needs an MI355X."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _grace(seed=0, n=1500, e=12000, d_in=64, h=32):
    from biomedkg_amd import synthetic
    from biomedkg_amd.gcl_module import GRACEModule
    torch.manual_seed(seed)
    mod = GRACEModule(in_dim=d_in, hidden_dim=h, out_dim=h, num_hidden_layers=2, fuse_method="attention").to(DEV).train()
    batch = SimpleNamespace(x=synthetic.lm_features(n, 2, d_in, seed=1).to(DEV), edge_index=synthetic.gcl_small(n, e, seed=2).to(DEV))
    return mod, batch


def test_replayed_grace_steps_equal_eager_steps_bit_for_bit():
    from biomedkg_amd import graphs as G, ops, rand as R
    from biomedkg_amd.optim import clip_adam_step
    ops.set_deterministic(True)      # bitwise claims need the fixed-order weight-gradient sums (Trainer(deterministic=True))
    n, e, d_in = 1500, 12000, 64
    g = torch.Generator(device=DEV).manual_seed(5)
    fm = [torch.rand(n, d_in, device=DEV, generator=g) >= 0.4 for _ in range(2)]
    ek = [torch.rand(e, device=DEV, generator=g) >= 0.4 for _ in range(2)]

    class Pinned(G._GraphRand):                       # the same two augmented views every step
        def __init__(self):
            super().__init__(seed=1)
            self.i = self.j = 0

        def masked(self, x, p):
            self.i += 1
            return x * fm[self.i % 2]

        def edge_keep(self, num_edges, p, device):
            self.j += 1
            return ek[self.j % 2]

    steps = 6
    # eager reference: warm-up (3) + steps eager steps with a capturable Adam (same arithmetic as the replayed one)
    mod, batch = _grace()
    mod.model.encoder.drop_out = False
    params = list(mod.model.parameters())
    opt = G.make_capturable(torch.optim.Adam(params, lr=1e-3, fused=True))
    eager_losses = []
    with R.use(Pinned()):
        for _ in range(3 + steps):
            opt.zero_grad(set_to_none=True)
            loss = mod.training_step(batch)
            loss.backward()
            clip_adam_step(opt, 1.0)                    # the step tail GraphedStep records (biomedkg_amd/optim.py)
            eager_losses.append(loss.detach().clone())
    want = [p.detach().clone() for p in params]

    mod2, batch2 = _grace()
    mod2.model.encoder.drop_out = False
    params2 = list(mod2.model.parameters())
    opt2 = torch.optim.Adam(params2, lr=1e-3, fused=True)
    step = G.GraphedStep(mod2, batch2, opt2, clip=1.0, warmup=3, provider=Pinned())
    try:
        got_losses = []
        for _ in range(steps):      # with unrelated allocations in between: a replay may only depend on what it owns
            junk = [torch.full((1 << 16,), float("nan"), device=DEV) for _ in range(32)]
            del junk
            got_losses.append(step().clone())
    finally:
        step.close()
    for a, b in zip(got_losses, eager_losses[3:]):
        assert torch.equal(a, b), (float(a), float(b))
    for a, b in zip(params2, want):
        assert torch.equal(a.detach(), b)


def test_replays_draw_fresh_randomness_and_train():
    from biomedkg_amd import graphs as G
    mod, batch = _grace(seed=3)
    params = list(mod.model.parameters())
    step = G.GraphedStep(mod, batch, torch.optim.Adam(params, lr=2e-3, fused=True), clip=1.0, warmup=2, seed=9)
    try:
        losses = [float(step()) for _ in range(40)]
    finally:
        step.close()
    assert all(l == l and abs(l) < 1e4 for l in losses)
    assert len(set(losses)) > 30                                   # new masks every replay
    assert sum(losses[-10:]) / 10 < sum(losses[:10]) / 10          # and the optimizer is really stepping


def test_in_kernel_dropout_follows_the_device_replay_epoch():
    """The library's Philox dropout under a captured launch: frozen (seed, stream id) arguments, yet every replay
    masks differently because the kernels add the device epoch; the masks equal the eager ones at the same epoch."""
    from biomedkg_amd import _lib as L, graph as Gr, ops
    torch.manual_seed(0)
    n, d = 512, 64
    ei = torch.randint(0, n, (2, 4000), device=DEV)
    gr = Gr.Graph(ei, n)
    gw = gr.gcn_weights()
    x = torch.rand(n, d, device=DEV) + 0.5
    c = gr.by_dst
    epoch = torch.zeros(1, dtype=torch.int32, device=DEV)
    run = lambda: ops.spmm(c.rowptr, c.col, gw.w_dst, x, n, self_w=gw.self_w, drop=ops.Dropout(0.3, seed=11, stream_id=4))
    L.call("bkg_set_rng_epoch", L.ptr(epoch))
    try:
        eager = []
        for k in range(1, 4):
            epoch.fill_(k)
            eager.append(run().clone())
        assert not torch.equal(eager[0] == 0, eager[1] == 0)
        epoch.zero_()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            epoch.add_(1)
            out = run()
        for k in range(3):
            graph.replay()
            assert torch.equal(out, eager[k])
    finally:
        L.call("bkg_set_rng_epoch", None)
    assert abs(float((eager[0] == 0).float().mean()) - 0.3) < 0.03


def test_weight_images_follow_replayed_updates():
    """bf16 mode caches a bf16 image of every weight per tensor VERSION; a replayed optimizer step changes the weights
    without touching the version.  Eager code after some replays (validation, export) must not read a stale image."""
    from biomedkg_amd import ops, rand as R
    from biomedkg_amd.graphs import GraphedStep
    ops.set_matmul_precision("bf16")
    try:
        mod, batch = _grace(seed=0, n=6000, e=60_000, d_in=256, h=256)       # tall enough for the streaming products
        opt = torch.optim.Adam(mod.model.parameters(), lr=1e-2)
        gs = GraphedStep(mod, batch, opt, seed=1)
        for _ in range(5):
            gs()
        torch.cuda.synchronize()

        def loss_now():
            torch.manual_seed(77)
            with torch.no_grad(), R.use(R.DeviceRand(seed=9)):
                return float(mod.training_step(batch))
        after = loss_now()
        for p in mod.model.parameters():             # what the loss is with every image rebuilt from the weights as they are
            p.__dict__.pop("_bkg_w16", None)
        fresh = loss_now()
        assert after == fresh, (after, fresh)
        # ... and again after MORE replays: the eager pass above left images cached under the version the replays keep
        for _ in range(5):
            gs()
        torch.cuda.synchronize()
        after = loss_now()
        for p in mod.model.parameters():
            p.__dict__.pop("_bkg_w16", None)
        fresh = loss_now()
        assert after == fresh, (after, fresh)
    finally:
        ops.set_matmul_precision("fp32")


def test_uncapturable_steps_are_refused():
    from biomedkg_amd import graphs as G, synthetic
    from biomedkg_amd.gcl_module import GGDModule
    torch.manual_seed(0)
    mod = GGDModule(in_dim=32, hidden_dim=16, out_dim=16, num_hidden_layers=1, fuse_method="none").to(DEV).train()
    batch = SimpleNamespace(x=torch.randn(400, 32, device=DEV), edge_index=synthetic.gcl_small(400, 3000, seed=1).to(DEV))
    with pytest.raises(G.CaptureRefused):
        G.GraphedStep(mod, batch, torch.optim.Adam(mod.model.parameters(), lr=1e-3, fused=True))


def test_no_replay_counter_is_left_registered_after_a_capture():
    """The library's in-kernel dropout streams read a replay counter through a registered device pointer; a captured
    step owns that counter.  Once the capture is done nothing outside the graph may launch with it: with the step object
    gone, two identical training-mode passes must draw identical masks (they read freed memory otherwise)."""
    import gc
    from biomedkg_amd import graphs as G, rand as R
    from biomedkg_amd.model.encoder import GCNEncoder
    mod, batch = _grace(seed=4)
    step = G.GraphedStep(mod, batch, torch.optim.Adam(mod.model.parameters(), lr=1e-3, fused=True), clip=1.0, warmup=2, seed=2)
    step()
    del step
    gc.collect()
    junk = [torch.randint(0, 1 << 30, (1 << 18,), device=DEV) for _ in range(64)]     # scribble over what was freed
    enc = GCNEncoder(64, 32, 32, 2).to(DEV).train()
    outs = []
    for _ in range(2):
        junk = [j + 1 for j in junk]
        with R.use(R.DeviceRand(seed=5)):
            outs.append(enc(batch.x[:, 0].contiguous(), batch.edge_index).detach().clone())
    assert torch.equal(outs[0], outs[1])


def test_every_fill_of_the_library_survives_replay():
    """A hipMemsetAsync recorded into a hipGraph re-executes wrongly on ROCm 7.2 (20 KB - 520 KB nodes leave stale
    words from the second replay on), so the library fills with kernels of its own.  gcn_norm (degree counters zeroed,
    then integer atomics) and the GraphSAINT sampler (node flags zeroed) are the two places where a stale fill shows
    at once: replayed three times each, they must reproduce the eager result."""
    from biomedkg_amd import synthetic
    from biomedkg_amd.graph import graph_of
    from biomedkg_amd.sampled import SaintDeviceSampler
    ei = synthetic.gcl_small(5000, 50_000, seed=42).to(DEV)
    gr = graph_of(ei, 5000)
    keep = torch.rand(ei.size(1), device=DEV) >= 0.4
    want = [t.clone() for t in gr.gcn_norm_raw(keep)]
    out = []
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = gr.gcn_norm_raw(keep)
    for _ in range(3):
        junk = [torch.full((1 << 16,), float("nan"), device=DEV) for _ in range(64)]     # whatever is free gets poisoned
        del junk
        g.replay()
        torch.cuda.synchronize()
        for a, b in zip(out, want):
            assert torch.equal(a, b)
    kg = synthetic.primekg_like(20000, 300000, 6, seed=2)
    a = SaintDeviceSampler(kg["edge_index"].to(DEV), kg["edge_type"].to(DEV), 20000, 64, seed=3, edge_cap=8192)
    b = SaintDeviceSampler(kg["edge_index"].to(DEV), kg["edge_type"].to(DEV), 20000, 64, seed=3, edge_cap=8192)
    for _ in range(200):
        a.draw(graph=True); b.draw(graph=False)
        assert a.sizes() == b.sizes()
    assert torch.equal(a.buf, b.buf)


def test_grace_replays_with_other_work_in_between():
    """Replays interleaved with unrelated allocations (a Trainer logging a loss every few steps): the learning rate is 0
    (the warm-up epoch of the reference's schedule), so the loss must stay where it started and nothing may turn NaN --
    it did, from the second replay on, while gcn_norm's counters were zeroed by a runtime memset node."""
    from biomedkg_amd import graphs as G
    mod, batch = _grace(seed=6)
    step = G.GraphedStep(mod, batch, torch.optim.Adam(mod.model.parameters(), lr=0.0, fused=True), clip=1.0, warmup=2, seed=4)
    before = [p.detach().clone() for p in mod.model.parameters()]
    losses = []
    for _ in range(12):
        junk = [torch.full((1 << 16,), float("nan"), device=DEV) for _ in range(64)]
        del junk
        losses.append(float(step()))
    assert all(v == v for v in losses) and max(losses) - min(losses) < 0.2, losses
    for a, p in zip(before, mod.model.parameters()):
        assert torch.equal(a, p.detach())


@pytest.mark.parametrize("n,hi", [(3000, 1409), (150_000, 27_000), (221_000, 27_000), (700_000, 1409), (1_500_000, 60_000),
                                  (3_000_000, 130_000), (8_000_000, 130_000)])
def test_captured_sorts_survive_replay(n, hi):
    """The library's key sort inside a captured step (the triplet orders and incidence lists of the padded KGE loop):
    replayed over changing keys it must stay a stable sort.  (The radix sort's one-sweep form zeroes histograms with
    runtime memsets -- broken as graph nodes on ROCm 7.2: a captured 1.5 M-key sort page-faulted on its second replay.
    Rounds 2-3 used the merge sort under capture; since round 4 rocPRIM's two memsets are routed to the library's fill
    kernel inside csrc/graph.hip and captured steps run the radix sort too.)"""
    from biomedkg_amd.graph import _Sorter
    keys = torch.randint(0, hi, (n,), device=DEV)

    def body():
        keys.add_(1).remainder_(hi)
        return _Sorter.sort(keys, None, hi, 1)
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sk, perm, ws = body()
    for _ in range(5):
        g.replay()
        torch.cuda.synchronize()
        ref = torch.sort(keys, stable=True)
        assert torch.equal(sk.long(), ref.values) and torch.equal(perm.long(), ref.indices)
