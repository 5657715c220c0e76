"""The reference's sampled GCL regime (NeighborLoader batches, ``biomedkg/data_module.py:93-99`` +
``gcl_module.py:52-56,229-234``) as padded, sync-free, hipGraph-replayed GGD steps (biomedkg_amd/sampled_gcl.py):

* the sync-free sampler hands out exactly the batch of the eager device sampler;
* ``bkg_randperm_dev`` permutes the real rows and leaves the pads in place, freshly per replay epoch;
* the padded step is the reference step on the real part of the batch: loss and every parameter gradient against the
  ordinary (unpadded) step on the same batch with the same draws, with and without GGD's augmentation branch, in both
  matmul precisions;
* a replayed graph is the same padded step run eagerly, bit for bit, over many steps, several edge buckets and both
  augmentation branches;
* the Trainer stand-in picks the loop for a GGD module over a NeighborBatchLoader.
Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _graph(n=3000, e=24000, seed=3):
    from biomedkg_amd import synthetic
    return synthetic.gcl_small(n, e, seed=seed).to(DEV), n


def _module(in_dim=64, hidden=32, seed=0, fuse=None, kind="ggd"):
    from biomedkg_amd import gcl_module as M
    torch.manual_seed(seed)
    cls = {"ggd": M.GGDModule, "dgi": M.DGIModule, "grace": M.GRACEModule}[kind]
    return cls(in_dim=in_dim, hidden_dim=hidden, out_dim=hidden, num_hidden_layers=2, learning_rate=1e-3,
               fuse_method=fuse).to(DEV)


def test_sync_free_sampler_equals_the_eager_device_sampler():
    from biomedkg_amd.sampled_gcl import NeighborDeviceSampler
    from biomedkg_amd.samplers import NeighborSampler
    ei, n = _graph()
    ref = NeighborSampler(ei, n, (5, 4, 3), seed=7)
    s = NeighborDeviceSampler(ei, n, (5, 4, 3), seed=7)
    for trial in range(3):
        seeds = torch.randint(0, n, (16,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(trial)).unique()
        n_id, sub, _ = ref.sample(seeds)
        s.draw(seeds)
        n_b, e_b = s.sizes()
        v = s.views(s.buf)
        assert n_b == n_id.numel() and e_b == sub.size(1) <= s.edge_cap
        assert torch.equal(v["n_id"][:n_b], n_id)
        assert torch.equal(v["src"][:e_b], sub[0]) and torch.equal(v["dst"][:e_b], sub[1])


def test_batches_list_their_edges_in_target_order():
    """What ``_bkg_dst_sorted`` vouches for: the device sampler expands frontier nodes in local-id order, hop after hop, so a
    batch's targets ascend; the padded batch keeps that (pads = self loops of the last node), and the target-side CSR built
    without a sort is the sorted one."""
    from biomedkg_amd import graph as G
    from biomedkg_amd.sampled_gcl import NeighborGraphedGcl
    from biomedkg_amd.samplers import NeighborSampler
    ei, n = _graph()
    s = NeighborSampler(ei, n, (5, 4, 3), seed=11)
    for trial in range(4):
        seeds = torch.randint(0, n, (24,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(trial)).unique()
        n_id, sub, _ = s.sample(seeds)
        assert getattr(sub, "_bkg_dst_sorted", None) == sub._version and bool((sub[1][1:] >= sub[1][:-1]).all())
        fast, slow = G.Graph(sub, n_id.numel()).by_dst, G.CSR(sub[1], sub[0], n_id.numel())
        assert torch.equal(fast.rowptr, slow.rowptr) and torch.equal(fast.col, slow.col) and torch.equal(fast.perm, slow.perm)
        shuffled = sub.clone()
        shuffled._bkg_dst_sorted = shuffled._version
        shuffled[:, :] = sub[:, torch.randperm(sub.size(1), device=DEV)]          # moved in place: the voucher is void
        again = G.Graph(shuffled, n_id.numel()).by_dst
        assert torch.equal(again.rowptr, slow.rowptr)
    x = torch.randn(n, 2, 64, device=DEV) * 0.3
    lp = NeighborGraphedGcl(_module(fuse="attention").train(), x, ei, batch_size=8, num_neighbors=(4, 3, 3), seed=5, graphs=False)
    lp.sampler.draw(torch.arange(8, device=DEV))
    n_b, e_b = lp.sampler.sizes()
    lp.cur.copy_(lp.sampler.buf)
    bucket = next(b for b in lp.buckets if b >= e_b)
    _, pei, _ = lp.padded_batch(bucket)
    assert pei._bkg_dst_sorted == pei._version and bool((pei[1][1:] >= pei[1][:-1]).all()) and bool((pei[:, e_b:] == n - 1).all())


def test_randperm_dev():
    from biomedkg_amd import _lib as L
    cap = 5000
    epoch = torch.zeros(1, dtype=torch.int32, device=DEV)
    seen = []
    for n in (0, 1, 2, 777, 4999, 5000, 9000):
        n_dev = torch.tensor([n], dtype=torch.int64, device=DEV)
        out = torch.empty(cap, dtype=torch.int64, device=DEV)
        L.call("bkg_randperm_dev", L.ptr(n_dev), cap, 12345, L.ptr(epoch), L.ptr(out), L.stream())
        m = min(n, cap)
        assert torch.equal(torch.sort(out[:m]).values, torch.arange(m, device=DEV))     # a permutation of the real rows
        assert torch.equal(out[m:], torch.arange(m, cap, device=DEV))                   # pads stay
        if n == 4999:
            seen.append(out.clone())
            epoch.add_(1)
            L.call("bkg_randperm_dev", L.ptr(n_dev), cap, 12345, L.ptr(epoch), L.ptr(out), L.stream())
            seen.append(out.clone())
            L.call("bkg_randperm_dev", L.ptr(n_dev), cap, 12346, L.ptr(epoch), L.ptr(out), L.stream())
            seen.append(out.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    assert int((seen[0][:4999] == torch.arange(4999, device=DEV)).sum()) < 50              # not the identity


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("aug", [False, True])
def test_padded_step_is_the_reference_step_on_the_real_part(precision, aug):
    from biomedkg_amd import rand as R, _lib as L, ops
    from biomedkg_amd.sampled_gcl import NeighborGraphedGcl
    ei, n = _graph()
    x = torch.randn(n, 2, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * 0.3
    prev = ops.get_matmul_precision()
    ops.set_matmul_precision(precision)
    try:
        mod = _module(fuse="attention").train()
        lp = NeighborGraphedGcl(mod, x, ei, batch_size=8, num_neighbors=(4, 3, 3), seed=5, graphs=False)
        seeds = torch.arange(40, 48, device=DEV)
        lp.sampler.draw(seeds)
        n_b, e_b = lp.sampler.sizes()
        lp.last_sizes = (n_b, e_b)
        lp.cur.copy_(lp.sampler.buf)
        assert 8 < n_b < n and 0 < e_b
        bucket = next(b for b in lp.buckets if b >= e_b)
        draws = {}

        class Padded(type(lp.rand)):                 # the padded step, its masks kept
            def feat_keep(self, x, p):
                draws["feat"] = super().feat_keep(x, p)
                return draws["feat"]

            def masked(self, x, p):
                return x * self.feat_keep(x, p)

            def edge_keep(self, num_edges, p, device):
                draws["edge"] = super().edge_keep(num_edges, p, device)
                return draws["edge"]

        class Same(Padded):                          # the ordinary step on the unpadded batch with the SAME draws
            def feat_keep(self, x, p):
                return draws["feat"][:x.size(0)]

            def edge_keep(self, num_edges, p, device):
                return draws["edge"][:num_edges].clone()

            def perm(self, n, device):
                self.counter += 1                    # (the padded provider's perm takes an in-kernel stream id too)
                return draws["perm"][:n].clone()

        L.call("bkg_set_rng_epoch", L.ptr(lp.epoch))
        try:
            padded = Padded(lp.rand.seed, lp.rand.n_dev, lp.rand.epoch)
            with R.use(padded):
                padded.begin_step(aug)
                mod.zero_grad(set_to_none=True)
                loss = lp.padded_loss(bucket)
                loss.backward()
                got = [p.grad.clone() for p in mod.model.parameters()]
            draws["perm"] = padded.last_perm
            assert torch.equal(torch.sort(padded.last_perm[:n_b]).values, torch.arange(n_b, device=DEV))
            assert ("feat" in draws) == aug
            same = Same(lp.rand.seed, lp.rand.n_dev, lp.rand.epoch)
            with R.use(same):
                same.begin_step(aug)
                mod.zero_grad(set_to_none=True)
                ref = mod.training_step(lp.unpadded_batch())
                ref.backward()
        finally:
            L.call("bkg_set_rng_epoch", None)
        tol = dict(rtol=2e-5, atol=1e-6) if precision == "fp32" else dict(rtol=2e-3, atol=2e-4)
        torch.testing.assert_close(loss, ref, **tol)
        gtol = dict(rtol=5e-4, atol=1e-6) if precision == "fp32" else dict(rtol=3e-2, atol=3e-4)
        for a, p in zip(got, mod.model.parameters()):
            torch.testing.assert_close(a, p.grad, **gtol)
    finally:
        ops.set_matmul_precision(prev)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("kind", ["dgi", "grace"])
def test_padded_dgi_and_grace_steps_are_the_reference_steps_on_the_real_part(kind, precision):
    """Round 4: DGI (masked readout + masked JSD) and GRACE (zeroed pad rows + the pad columns' exp(0) taken out of every
    denominator) in the padded form equal ``DGIModule`` / ``GRACEModule.training_step`` (reference gcl_module.py:140-143,
    186-190) on the unpadded batch with the SAME draws: loss and every parameter gradient."""
    from biomedkg_amd import rand as R, _lib as L, ops
    from biomedkg_amd.sampled_gcl import NeighborGraphedGcl
    ei, n = _graph()
    x = torch.randn(n, 2, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * 0.3
    prev = ops.get_matmul_precision()
    ops.set_matmul_precision(precision)
    try:
        mod = _module(fuse="attention", kind=kind).train()
        lp = NeighborGraphedGcl(mod, x, ei, batch_size=8, num_neighbors=(4, 3, 3), seed=5, graphs=False)
        seeds = torch.arange(40, 48, device=DEV)
        lp.sampler.draw(seeds)
        n_b, e_b = lp.sampler.sizes()
        lp.last_sizes = (n_b, e_b)
        lp.cur.copy_(lp.sampler.buf)
        assert 8 < n_b < n and 0 < e_b
        bucket = next(b for b in lp.buckets if b >= e_b)
        tape = {"feat": [], "edge": [], "perm": []}

        class Padded(type(lp.rand)):                 # the padded step, its draws kept in order
            def feat_keep(self, x, p):
                tape["feat"].append(super().feat_keep(x, p))
                return tape["feat"][-1]

            def masked(self, x, p):
                return x * self.feat_keep(x, p)

            def edge_keep(self, num_edges, p, device):
                tape["edge"].append(super().edge_keep(num_edges, p, device))
                return tape["edge"][-1]

            def perm(self, n, device):
                tape["perm"].append(super().perm(n, device))
                return tape["perm"][-1]

        class Same(Padded):                          # the ordinary step on the unpadded batch with the SAME draws
            def feat_keep(self, x, p):
                return tape["feat"].pop(0)[:x.size(0)]

            def edge_keep(self, num_edges, p, device):
                return tape["edge"].pop(0)[:num_edges].clone()

            def perm(self, n, device):
                self.counter += 1
                return tape["perm"].pop(0)[:n].clone()

        L.call("bkg_set_rng_epoch", L.ptr(lp.epoch))
        try:
            padded = Padded(lp.rand.seed, lp.rand.n_dev, lp.rand.epoch)
            with R.use(padded):
                padded.begin_step(False)
                mod.zero_grad(set_to_none=True)
                # GRACE's NT-Xent runs over a rung of the node ladder (the first n_pad >= n_b rows) instead of all n_cap
                n_pad = -(-n_b // 16) * 16 if kind == "grace" else None
                assert n_pad is None or n_b <= n_pad < lp.n_cap
                loss = lp.padded_loss(bucket, n_pad)
                loss.backward()
                got = [p.grad.clone() for p in mod.model.parameters()]
                if kind == "grace":                  # ... and gives what the whole capacity gives (pads add nothing)
                    mod.zero_grad(set_to_none=True)
                    tape_keep = {k: list(v) for k, v in tape.items()}
                    replay = Same(lp.rand.seed, lp.rand.n_dev, lp.rand.epoch)
                    replay.feat_keep = lambda x, p: tape_keep["feat"].pop(0)
                    replay.edge_keep = lambda num_edges, p, device: tape_keep["edge"].pop(0)
                    with R.use(replay):
                        replay.begin_step(False)
                        whole = lp.padded_loss(bucket, None)
                    torch.testing.assert_close(whole, loss, rtol=2e-5 if precision == "fp32" else 3e-3, atol=1e-6)
                    mod.zero_grad(set_to_none=True)
            assert len(tape["perm"]) == (1 if kind == "dgi" else 0) and len(tape["feat"]) == (2 if kind == "grace" else 0)
            same = Same(lp.rand.seed, lp.rand.n_dev, lp.rand.epoch)
            with R.use(same):
                same.begin_step(False)
                mod.zero_grad(set_to_none=True)
                ref = mod.training_step(lp.unpadded_batch())
                ref.backward()
        finally:
            L.call("bkg_set_rng_epoch", None)
        assert torch.isfinite(loss) and all(torch.isfinite(g).all() for g in got)
        tol = dict(rtol=2e-5, atol=1e-6) if precision == "fp32" else dict(rtol=3e-3, atol=3e-4)
        torch.testing.assert_close(loss, ref, **tol)
        for a, p in zip(got, mod.model.parameters()):
            err = float((a - p.grad).norm() / p.grad.norm().clamp_min(1e-12))
            assert err < (5e-4 if precision == "fp32" else 5e-2), err
    finally:
        ops.set_matmul_precision(prev)


def _run(graphs, steps, seed=0, precision="fp32", max_graphs=16, kind="ggd"):
    from biomedkg_amd import ops
    from biomedkg_amd.sampled_gcl import NeighborGraphedGcl
    ei, n = _graph()
    x = torch.randn(n, 2, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * 0.3
    prev = ops.get_matmul_precision()
    ops.set_matmul_precision(precision)
    ops.set_deterministic(True)      # bitwise claims need the fixed-order weight-gradient sums (Trainer(deterministic=True))
    try:
        mod = _module(fuse="attention", kind=kind).train()
        lp = NeighborGraphedGcl(mod, x, ei, batch_size=6, num_neighbors=(4, 3, 2), seed=seed, graphs=graphs,
                                buckets=(64, 128, 192, 256, 384, 512, 1024, 24000), max_graphs=max_graphs)
        losses, sizes = [], []
        for _ in range(steps):
            losses.append(lp.step().clone())
            sizes.append(lp.last_sizes + (lp.last_aug,))
        torch.cuda.synchronize()
        lp.close()
        return torch.stack(losses), sizes, lp.stats, [p.detach().clone() for p in mod.model.parameters()]
    finally:
        ops.set_matmul_precision(prev)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_replayed_steps_equal_eager_padded_steps(precision):
    steps = 40
    la, sa, stats_a, pa = _run(True, steps, precision=precision)
    lb, sb, stats_b, pb = _run(False, steps, precision=precision)
    assert sa == sb                                                      # the same batches and coins
    assert len({s[:2] for s in sa}) > 5 and {s[2] for s in sa} == {False, True}
    assert stats_a["captured"] >= 3 and stats_a["replayed"] >= steps // 2 and stats_b["eager_padded"] == steps
    assert torch.isfinite(la).all()
    assert torch.equal(la, lb)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
    assert float(la[-5:].mean()) < float(la[:5].mean())                  # it trains


@pytest.mark.parametrize("kind", ["dgi", "grace"])
def test_replayed_dgi_and_grace_steps_equal_eager_padded_steps(kind):
    steps = 30
    la, sa, stats_a, pa = _run(True, steps, kind=kind)
    lb, sb, stats_b, pb = _run(False, steps, kind=kind)
    assert sa == sb and len({s[:2] for s in sa}) > 5
    assert stats_a["captured"] >= 2 and stats_a["replayed"] >= steps // 3 and stats_b["eager_padded"] == steps
    assert torch.isfinite(la).all()
    assert torch.equal(la, lb)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


def test_graph_count_is_capped():
    """Past ``max_graphs`` captured (bucket, branch) pairs the remaining kinds of batch run the same padded step eagerly."""
    la, sa, stats_a, pa = _run(True, 30, max_graphs=2)
    lb, sb, stats_b, pb = _run(False, 30)
    assert stats_a["captured"] == 2 and stats_a["eager_padded"] > 2 and stats_a["replayed"] > 0
    assert torch.equal(la, lb) and all(torch.equal(a, b) for a, b in zip(pa, pb))


def test_trainer_picks_the_graphed_loop():
    from biomedkg_amd.samplers import NeighborBatchLoader
    from biomedkg_amd.trainer import Trainer
    ei, n = _graph(n=1200, e=9000)
    x = torch.randn(n, 2, 64, device=DEV) * 0.3
    mod = _module(fuse="attention")
    loader = NeighborBatchLoader(x, ei, None, 64, (4, 3, 2), True, seed=3)
    tr = Trainer(max_epochs=2, devices=1, enable_progress_bar=False)
    step = tr._try_graph(mod.to(DEV).train(), loader, torch.optim.Adam(mod.model.parameters(), lr=1e-3, fused=True))
    assert step is not None and step.owns_batches and type(step.loop).__name__ == "NeighborGraphedGcl"
    step.loop.close()
    tr.fit(mod, loader)
    assert tr.global_step == 2 * len(loader)
    assert all(torch.isfinite(p).all() for p in mod.parameters())
