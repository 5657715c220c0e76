"""The reference's sampled KGE regime (GraphSAINT random-walk batches, ``biomedkg/data_module.py:100-107`` +
``kge_module.py:104-131``) as padded, sync-free, hipGraph-replayed steps (biomedkg_amd/sampled.py):

* the sync-free sampler hands out exactly the batch of the eager device sampler (same walks -> same node set, same
  induced edges, same relabelling), padded as documented, with the sizes in device memory;
* the device-extent negative sampler keeps PyG's contract (k distinct off-diagonal non-edges among the REAL nodes,
  relation ids = a permutation of the positives' ids repeated neg_ratio times) and pads the rest;
* the padded step is the reference step on the real part of the batch: loss and every parameter gradient against the
  ordinary (unpadded) step on the same batch with the same negatives;
* a replayed graph is the same padded step run eagerly, bit for bit, over many steps and several edge buckets;
* batches outside the captured regime take the ordinary step.
Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _graph(n=3000, e=40000, r=6, seed=3):
    from biomedkg_amd import synthetic
    g = synthetic.primekg_like(n, e, r, seed=seed)
    return g["edge_index"].to(DEV), g["edge_type"].to(DEV), n, g["num_relations"]


def _module(r, dim=64, decoder="dismult", neg_ratio=3, seed=0, layers=1):
    from biomedkg_amd.kge_module import KGEModule
    torch.manual_seed(seed)
    return KGEModule(encoder_name="rgcn", decoder_name=decoder, in_dim=dim, hidden_dim=dim, out_dim=dim,
                     num_hidden_layers=layers, num_relation=r, num_heads=2, scheduler_type="cosine", learning_rate=1e-3,
                     warm_up_ratio=0.0, fuse_method="none", neg_ratio=neg_ratio, node_init_method="random").to(DEV)


def test_sync_free_sampler_equals_the_eager_device_sampler():
    from biomedkg_amd.sampled import SaintDeviceSampler
    from biomedkg_amd.samplers import SaintRandomWalkSampler
    ei, et, n, _ = _graph()
    walks, cap = 48, 4096
    for trial in range(3):
        start = torch.randint(0, n, (walks,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(trial))
        ref = SaintRandomWalkSampler(ei, n, walks, walk_length=10, seed=7)
        n_id, sub, e_id = ref.sample_with_torch_glue(start)                 # its walk seed: 7 * 1_000_003 + 1
        s = SaintDeviceSampler(ei, et, n, walks, walk_length=10, seed=7 * 1_000_003 + 1, edge_cap=cap, pad_types=5)
        s.draw(start=start, use_epoch=False)
        n_b, e_b, e_free = s.sizes()
        v = s.views(s.buf, s.node_cap, s.edge_cap)
        assert n_b == n_id.numel() and e_b == e_id.numel() <= cap
        assert torch.equal(v["n_id"][:n_b], n_id) and bool((v["n_id"][n_b:] == n).all())
        assert torch.equal(v["src"][:e_b], sub[0]) and torch.equal(v["dst"][:e_b], sub[1])
        assert torch.equal(v["eid"][:e_b], e_id) and torch.equal(v["type"][:e_b], et[e_id])
        pads = torch.arange(e_b, cap, device=DEV)            # pad edges: into the dummy node, sources / types spread
        assert torch.equal(v["src"][e_b:], pads % s.node_cap) and bool((v["dst"][e_b:] == s.pad_node).all())
        assert bool((v["eid"][e_b:] == -1).all()) and torch.equal(v["type"][e_b:], pads % 5)
        assert e_free == int((sub[0] != sub[1]).sum())
        assert n_b <= s.node_cap - 1                                        # the dummy node is always a pad


def test_sampler_roots_and_replays():
    """No roots given: they come from the keyed stream; a captured draw replays to fresh batches, the same ones an
    eager sampler with the same seed draws."""
    from biomedkg_amd.sampled import SaintDeviceSampler
    ei, et, n, _ = _graph()
    a = SaintDeviceSampler(ei, et, n, 32, seed=11, edge_cap=4096)
    b = SaintDeviceSampler(ei, et, n, 32, seed=11, edge_cap=4096)
    seen = []
    for _ in range(4):
        a.draw(graph=True); b.draw(graph=False)
        assert a.sizes() == b.sizes()
        assert torch.equal(a.buf, b.buf)
        seen.append(a.buf.clone())
        assert 0 <= int(a.walks.min()) and int(a.walks.max()) < n
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


def _check_negatives(sg, e_b, bucket):
    neg, neg_type, count = sg.last_negatives
    n_b, e_real = int(sg.v["sizes"][0]), int(sg.v["sizes"][1])
    k = sg.neg_ratio * e_real
    assert int(count) == k and neg.size(1) == sg.neg_ratio * bucket
    h, t = neg[0, :k], neg[1, :k]
    assert bool((h != t).all()) and int(h.min()) >= 0 and int(t.min()) >= 0 and int(h.max()) < n_b and int(t.max()) < n_b
    code = h * n_b + t
    assert torch.unique(code).numel() == k                                   # distinct (random.sample's guarantee)
    pos = sg.v["src"][:e_real] * n_b + sg.v["dst"][:e_real]
    assert not bool(torch.isin(code, pos).any())                             # none is an edge of the batch
    pads = neg[:, k:]                                                        # spread over the real nodes, never scored
    assert int(pads.min()) >= 0 and int(pads.max()) < n_b and bool((neg_type[k:] == sg.num_relations).all())
    want = torch.sort(sg.v["type"][:e_real].repeat(sg.neg_ratio)).values     # a permutation of the repeated ids
    assert torch.equal(torch.sort(neg_type[:k]).values, want)


@pytest.mark.parametrize("decoder", ["dismult", "transe"])
def test_padded_step_is_the_reference_step_on_the_real_part(decoder):
    from biomedkg_amd import rand as R, _lib as L
    from biomedkg_amd.sampled import SaintGraphedKge
    ei, et, n, r = _graph()
    x = torch.randn(n, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * 0.3
    for train_mode in (False, True):
        mod = _module(r, decoder=decoder)
        mod.train(train_mode)
        sg = SaintGraphedKge(mod, x, ei, et, batch_size=40, seed=5, graphs=False, buckets=(512, 1024, 2048, 4096, 8192))
        sg.sampler.draw(graph=False)
        n_b, e_b, e_free = sg.sampler.sizes()
        sg.cur.copy_(sg.sampler.buf)
        bucket = sg.bucket_of(n_b, e_b, e_free)
        assert bucket is not None and bucket >= e_b
        L.call("bkg_set_rng_epoch", L.ptr(sg.epoch))
        try:
            with R.use(sg.rand):
                sg.rand.begin_step()
                mod.zero_grad(set_to_none=True)
                loss = sg.padded_loss(bucket)
                loss.backward()
                got = [p.grad.clone() for p in mod.model.parameters()]
                _check_negatives(sg, e_b, bucket)
                neg, neg_type, count = sg.last_negatives
                k = int(count)

                class Same(type(sg.rand)):                                   # the ordinary step with the SAME draws
                    def neg_edges(self, edge_index, num_neg_samples):
                        return neg[:, :k].contiguous()

                    def shuffle(self, src):
                        return neg_type[:k].contiguous()

                    def shuffle_repeat(self, src, repeat):
                        return neg_type[:k].contiguous()

                    def grouped_types(self, *a):             # (the padded step's own ids, not a grouped list of its own)
                        return None
                same = Same(sg.rand.seed)
                with R.use(same):
                    same.begin_step()
                    mod.zero_grad(set_to_none=True)
                    ref = mod.training_step(sg.unpadded_batch(n_b, e_b))
                    ref.backward()
        finally:
            L.call("bkg_set_rng_epoch", None)
        torch.testing.assert_close(loss, ref, rtol=2e-6, atol=1e-7)
        for a, p in zip(got, mod.model.parameters()):
            torch.testing.assert_close(a, p.grad, rtol=2e-4, atol=2e-7)


def _run(graphs, steps, buckets, seed=0, neg_ratio=3, walks=40):
    from biomedkg_amd.sampled import SaintGraphedKge
    ei, et, n, r = _graph()
    x = torch.randn(n, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * 0.3
    mod = _module(r, neg_ratio=neg_ratio, seed=seed, layers=1).train()
    sg = SaintGraphedKge(mod, x, ei, et, batch_size=walks, seed=5, graphs=graphs, buckets=buckets)
    losses, sizes = [], []
    for _ in range(steps):
        junk = [torch.full((1 << 16,), float("nan"), device=DEV) for _ in range(16)]   # (a replay may only depend on
        del junk                                                                      #  what it owns)
        losses.append(sg.step().clone())
        sizes.append(sg.last_sizes)
    return sg, losses, sizes, [p.detach().clone() for p in mod.model.parameters()]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_replayed_steps_equal_the_eager_padded_steps_bit_for_bit(precision):
    from biomedkg_amd import ops
    ops.set_deterministic(True)      # bitwise claims need the fixed-order weight-gradient sums (Trainer(deterministic=True))
    ops.set_matmul_precision(precision)
    buckets = (256, 384, 512, 768, 1024, 1536, 2048, 4096, 8192)            # narrow ladder: several buckets get used
    a, la, sa, pa = _run(True, 40, buckets)
    b, lb, sb, pb = _run(False, 40, buckets)
    assert sa == sb                                                          # the same batches
    assert a.stats["captured"] >= 2 and a.stats["replayed"] >= 25, a.stats
    assert b.stats["eager_padded"] == 40 - b.stats["eager_fallback"]
    for i, (u, v) in enumerate(zip(la, lb)):
        assert torch.equal(u, v), (i, float(u), float(v))
    for u, v in zip(pa, pb):
        assert torch.equal(u, v)
    assert float(la[-1]) < float(la[0])                                      # and it trains


def test_batches_outside_the_ladder_take_the_ordinary_step():
    sg, losses, sizes, _ = _run(True, 6, buckets=(64, 128))                  # no batch fits: every step is the eager one
    assert sg.stats["eager_fallback"] == 6 and sg.stats["captured"] == 0
    assert all(torch.isfinite(l) for l in losses)


def test_over_capacity_batches_are_not_truncated():
    """A batch with more induced edges than the sampler's edge capacity (= the largest bucket): the sampler stores the
    first ``edge_cap`` edges and reports the true total; the fallback must train on ALL of them -- the batch the eager
    sampler builds from the same node set -- and its loss must be the ordinary step's loss on that batch."""
    from biomedkg_amd import rand as R
    from biomedkg_amd.sampled import SaintGraphedKge
    from biomedkg_amd.samplers import _relabel
    ei, et, n, r = _graph()
    x = torch.randn(n, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * 0.3
    mod = _module(r, neg_ratio=2, layers=1).eval()             # eval: no dropout stream to line up
    sg = SaintGraphedKge(mod, x, ei, et, batch_size=40, seed=5, graphs=True, buckets=(64, 128))
    sg.step()
    n_b, e_b, _ = sg.last_sizes
    assert e_b > sg.e_cap                                       # the case: the buffers hold 128 of e_b edges
    b = sg.unpadded_batch(n_b, e_b)
    n_id = sg.v["n_id"][:n_b]
    mask = torch.zeros(n, dtype=torch.bool, device=DEV); mask[n_id] = True
    e_id = torch.nonzero(mask[ei[0]] & mask[ei[1]]).squeeze(1)
    src, dst = _relabel(n_id, n, ei[0][e_id], ei[1][e_id])
    assert b.edge_index.size(1) == e_b == e_id.numel()
    assert torch.equal(b.edge_index, torch.stack([src, dst])) and torch.equal(b.edge_type, et[e_id])
    assert torch.equal(b.e_id, e_id) and torch.equal(b.n_id, n_id)
    assert torch.equal(b.x, x[n_id])
    # the first 128 stored edges are a prefix of the full list (what the truncated step used to train on)
    assert torch.equal(sg.v["src"][:128], src[:128]) and torch.equal(sg.v["dst"][:128], dst[:128])


@pytest.mark.parametrize("init,fuse,decoder,fix", [("lm", "attention", "dismult", None), ("lm", "redaf", "transe", None),
                                                   ("gcl", "none", "dismult", 1), ("random", "none", "transe", None)])
def test_every_input_form_of_the_kge_module_replays(init, fuse, decoder, fix):
    """[N, 2, D] language-model features through the attention / ReDAF fuser, [N, 1, D] GCL embeddings, fix_edge_id
    (the DPI fine-tuning setting), TransE: the padded step captures and replays, and tracks the eager padded loop."""
    from biomedkg_amd import synthetic
    from biomedkg_amd.kge_module import KGEModule
    from biomedkg_amd.sampled import SaintGraphedKge
    ei, et, n, r = _graph()
    gen = torch.Generator(device=DEV).manual_seed(2)
    if init == "lm":
        x = synthetic.lm_features(n, 2, 64, seed=4).to(DEV)
    elif init == "gcl":
        x = torch.randn(n, 1, 64, device=DEV, generator=gen) * 0.3
    else:
        x = torch.randn(n, 64, device=DEV, generator=gen) * 0.3
    runs = {}
    for graphs in (True, False):
        torch.manual_seed(0)
        mod = KGEModule(encoder_name="rgcn", decoder_name=decoder, in_dim=64, hidden_dim=64, out_dim=64, num_hidden_layers=1,
                        num_relation=r, num_heads=2, scheduler_type="cosine", learning_rate=1e-3, warm_up_ratio=0.0,
                        fuse_method=fuse, neg_ratio=2, node_init_method=init).to(DEV).train()
        if fix is not None:
            mod.fix_edge_id = fix
        sg = SaintGraphedKge(mod, x, ei, et, batch_size=32, seed=9, graphs=graphs, buckets=(512, 1024, 2048, 4096))
        runs[graphs] = [float(sg.step()) for _ in range(12)]
        if graphs:
            assert sg.graphs_on and sg.stats["replayed"] >= 6, sg.stats
    a, b = torch.tensor(runs[True]), torch.tensor(runs[False])
    assert torch.isfinite(a).all()
    torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-5)     # (float atomics in the weight gradients: not bitwise here)


def test_the_epoch_scheduler_reaches_the_replayed_steps():
    """The optimizer's learning rate is a device scalar (capturable Adam): torch's schedulers fill it in place, so a
    replayed graph trains with the scheduled rate -- here a schedule that drops it to 0 freezes the weights."""
    from biomedkg_amd.sampled import SaintGraphedKge
    ei, et, n, r = _graph()
    x = torch.randn(n, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * 0.3
    mod = _module(r).train()
    opt = torch.optim.Adam(mod.model.parameters(), lr=5e-3, fused=True)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda epoch: 1.0 if epoch == 0 else 0.0)   # built on the float rate
    sg = SaintGraphedKge(mod, x, ei, et, batch_size=40, optimizer=opt, seed=5, buckets=(4096,))
    for _ in range(6):
        sg.step()
    assert sg.stats["replayed"] >= 4
    before = [p.detach().clone() for p in mod.model.parameters()]
    sg.step()
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, mod.model.parameters()))   # still training
    sched.step()                                                                                 # "epoch 1": rate 0
    assert float(opt.param_groups[0]["lr"]) == 0.0 and isinstance(opt.param_groups[0]["lr"], torch.Tensor)
    before = [p.detach().clone() for p in mod.model.parameters()]
    for _ in range(3):
        sg.step()
    assert sg.stats["captured"] == 1
    for a, p in zip(before, mod.model.parameters()):
        assert torch.equal(a, p.detach())


def test_fusing_the_table_once_equals_fusing_every_batch():
    """The device loaders tag ``x = table[n_id]``; AttentionFusion (row-wise, never trained) then fuses the table once
    and gathers: bit for bit what fusing the gathered rows gives."""
    from biomedkg_amd import synthetic
    from biomedkg_amd.samplers import rows_of
    from biomedkg_amd.utils.fusion import AttentionFusion
    torch.manual_seed(0)
    fuse = AttentionFusion(64).to(DEV)
    table = synthetic.lm_features(6000, 2, 64, seed=1).to(DEV)
    for seed in (1, 2):
        n_id = torch.randint(0, 6000, (1500,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed))
        tagged = rows_of(table, n_id)
        plain = table[n_id]
        assert torch.equal(tagged, plain)
        a, b = fuse(tagged), fuse(plain)
        assert torch.equal(a, b)
    assert any(h[2] is table for h in fuse.__dict__["_fused_cache"])


def test_fusion_cache_is_not_fooled_by_a_recycled_address():
    """A mini-batch loader's next batch usually lands on the address of the one just freed: the cached AttentionFusion
    output must belong to the tensor OBJECT, not to its address."""
    from biomedkg_amd import synthetic
    from biomedkg_amd.utils.fusion import AttentionFusion
    torch.manual_seed(0)
    fuse = AttentionFusion(64).to(DEV)
    a = synthetic.lm_features(500, 2, 64, seed=1).to(DEV)
    out_a = fuse(a).clone()
    del a                                                   # (b usually lands on a's address)
    b = synthetic.lm_features(500, 2, 64, seed=2).to(DEV)
    out_b = fuse(b)
    fresh = AttentionFusion(64).to(DEV)
    fresh.load_state_dict(fuse.state_dict())
    torch.testing.assert_close(out_b, fresh(b), rtol=0, atol=0)
    assert not torch.equal(out_b, out_a)


def test_trainer_replays_the_saint_regime(tmp_path):
    """train_kge.py with data.sampler=device: the Trainer stand-in hands the GraphSAINT loader to the graphed loop."""
    import train_kge
    from biomedkg_amd import trainer as T
    seen = {}
    orig = T.Trainer._try_graph

    def spy(self, model, loader, opt):
        step = orig(self, model, loader, opt)
        seen["step"] = step
        return step
    T.Trainer._try_graph = spy
    try:
        train_kge.main(["devices=[0]", "data=synthetic", "data.sampler=device", "data.num_nodes=2000", "data.num_edges=20000",
                        "data.embed_dim=64", "data.batch_size=32", "data.steps_per_epoch=12", "model.in_dim=64",
                        "model.hidden_dim=64", "model.out_dim=64", "model.num_hidden_layers=1", "epochs=2",
                        "val_every_epoch=1", "neg_ratio=2", f"ckpt_dir={tmp_path}", f"log_dir={tmp_path}"])
    finally:
        T.Trainer._try_graph = orig
    st = seen["step"].loop.stats
    assert st["replayed"] + st["captured"] + st["eager_fallback"] == 24 and st["replayed"] > 0, st


def test_three_thousand_replays():
    """A soak: the failure this guards against (a captured SAMPLER graph page-faulting after a few thousand replays on
    ROCm 7.2) only showed between 1,800 and 4,000 steps of an end-to-end run; the loop as built launches the sampler
    eagerly and replays only the step."""
    from biomedkg_amd.sampled import SaintGraphedKge
    ei, et, n, r = _graph(n=20000, e=400000, r=10)
    x = torch.randn(n, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * 0.1
    mod = _module(r, neg_ratio=5).train()
    sg = SaintGraphedKge(mod, x, ei, et, batch_size=64, seed=5)
    first = float(sg.step())
    for _ in range(3000):
        loss = sg.step()
    sg.close()
    assert sg.stats["replayed"] >= 2900 and torch.isfinite(loss) and float(loss) < first
    assert all(bool(torch.isfinite(p).all()) for p in mod.model.parameters())
