"""Device samplers: structural properties of PyG's NeighborLoader / GraphSAINT-RW semantics, uniformity of the
draws, and an end-to-end training step on sampled batches.  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _graph(n=500, e=20_000, seed=0):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    ei[1, :3000] = 7                       # a hub with ~3000 in-edges
    ei[0, ei[0] == 13] = 14                # node 13: no out-edges (walks must stay put)
    return ei


def test_neighbor_sampler_properties():
    from biomedkg_amd.samplers import NeighborSampler
    n = 500
    ei = _graph(n)
    samp = NeighborSampler(ei.to(DEV), n, (30, 30, 30), seed=1)
    seeds = torch.tensor([7, 3, 99, 250], device=DEV)
    n_id, sub, e_id = samp.sample(seeds)
    n_id, sub, e_id = n_id.cpu(), sub.cpu(), e_id.cpu()
    assert n_id[:4].tolist() == [7, 3, 99, 250] and n_id.unique().numel() == n_id.numel()   # seeds first, no repeats
    assert torch.equal(n_id[sub[0]], ei[0, e_id]) and torch.equal(n_id[sub[1]], ei[1, e_id])  # real edges, relabelled
    assert e_id.unique().numel() == e_id.numel()                                              # without replacement
    indeg = torch.bincount(ei[1], minlength=n)
    per_dst = torch.bincount(n_id[sub[1]], minlength=n)
    expanded = per_dst > 0
    assert (per_dst[expanded] == torch.clamp(indeg[expanded], max=30)).all()    # min(deg, 30) per expanded node
    assert per_dst[7] == 30 and indeg[7] > 2500
    # 1-hop with fanout -1-like (64 cap) takes everything for small degrees
    full = NeighborSampler(ei.to(DEV), n, (64,), seed=2)
    _, sub1, e1 = full.sample(torch.tensor([3], device=DEV))
    assert e1.numel() == min(int(indeg[3]), 64)


def test_neighbor_sampler_is_uniform_over_the_hub_edges():
    from biomedkg_amd.samplers import NeighborSampler
    n = 500
    ei = _graph(n)
    samp = NeighborSampler(ei.to(DEV), n, (30,), seed=3)
    hub_edges = torch.nonzero(ei[1] == 7).squeeze(1)
    counts = torch.zeros(ei.size(1))
    trials = 400
    for _ in range(trials):
        _, _, e_id = samp.sample(torch.tensor([7], device=DEV))
        counts[e_id.cpu()] += 1
    c = counts[hub_edges]
    expect = trials * 30 / hub_edges.numel()
    assert abs(c.mean() - expect) < 1e-6 and c.max() < expect + 6 * expect ** 0.5 + 3 and c.min() >= 0
    assert (c > 0).float().mean() > 0.9                      # every hub edge gets picked sooner or later


def test_saint_random_walk_sampler_properties():
    from biomedkg_amd.samplers import SaintRandomWalkSampler
    n = 500
    ei = _graph(n, seed=4)
    samp = SaintRandomWalkSampler(ei.to(DEV), n, batch_size=64, walk_length=10, seed=5)
    start = torch.randint(0, n, (64,), device=DEV)
    start[0] = 13
    walks = samp.walks(start).cpu()
    assert walks.shape == (64, 11) and torch.equal(walks[:, 0], start.cpu())
    adj = set(map(tuple, ei.t().tolist()))
    has_out = torch.bincount(ei[0], minlength=n) > 0
    for w in walks.tolist():
        for a, b in zip(w[:-1], w[1:]):
            assert (a, b) in adj or (a == b and not has_out[a])
    assert (walks[0] == 13).all()                             # a sink stays put
    n_id, sub, e_id = samp.sample(start)
    n_id, sub, e_id = n_id.cpu(), sub.cpu(), e_id.cpu()
    # the node set: sorted, distinct, contains every start node, and every node in it is reachable from a start node within
    # the walk length (the walks themselves are re-drawn per call -- a new call id -- so they cannot be compared one to one;
    # test_saint_random_walk_sampler_equals_the_oracle_on_the_same_tape below pins them against the oracle)
    assert torch.equal(n_id, torch.unique(n_id)) and bool(torch.isin(start.cpu(), n_id).all())
    reach = torch.zeros(n, dtype=torch.bool); reach[start.cpu()] = True
    for _ in range(10):
        step = torch.zeros(n, dtype=torch.bool); step[ei[1][reach[ei[0]]]] = True
        reach |= step
    assert bool(reach[n_id].all())
    inside = torch.zeros(n, dtype=torch.bool); inside[n_id] = True
    expect = torch.nonzero(inside[ei[0]] & inside[ei[1]]).squeeze(1)
    assert torch.equal(e_id, expect)                          # the induced subgraph: ALL edges among the nodes
    assert torch.equal(n_id[sub[0]], ei[0, e_id]) and torch.equal(n_id[sub[1]], ei[1, e_id])


def test_training_steps_on_sampled_batches():
    from biomedkg_amd import synthetic
    from biomedkg_amd.gcl_module import GRACEModule
    from biomedkg_amd.kge_module import KGEModule
    from biomedkg_amd.samplers import NeighborBatchLoader, SaintBatchLoader
    g = synthetic.primekg_like(3000, 60_000, 8, seed=6)
    ei, et = g["edge_index"].to(DEV), g["edge_type"].to(DEV)
    x = torch.randn(3000, 32, device=DEV)
    gcl = GRACEModule(in_dim=32, hidden_dim=16, out_dim=16, num_hidden_layers=2, fuse_method="none").to(DEV).train()
    for i, b in enumerate(NeighborBatchLoader(x, ei, None, batch_size=64, seed=1)):
        loss = gcl.training_step(b)
        loss.backward()
        assert torch.isfinite(loss) and b.batch_size == 64 and b.x.size(0) == b.n_id.numel()
        if i == 2:
            break
    kge = KGEModule("rgcn", "dismult", 32, 16, 16, 2, g["num_relations"], 2, "cosine", 1e-3, 0.2, "none", 1, "random").to(DEV).train()
    loader = SaintBatchLoader(x, ei, et, batch_size=64, walk_length=10, num_steps=3, seed=2)
    assert len(loader) == 3
    for b in loader:
        loss = kge.training_step(b)
        loss.backward()
        assert torch.isfinite(loss) and b.edge_type.numel() == b.edge_index.size(1) and b.x.size(0) <= 64 * 11


def test_one_read_neighbor_sampler_equals_the_hop_by_hop_sampler():
    """The same batch draw for draw: node order (seeds, then each hop's new nodes ascending), edge order ((frontier row,
    pick) per hop), original edge ids -- over several seed sets, fan-outs (incl. more than any degree) and graphs with
    isolated nodes."""
    from biomedkg_amd import synthetic
    from biomedkg_amd.samplers import NeighborSampler
    for n, e, fans in ((3000, 40000, (30, 30, 30)), (500, 900, (2, 5)), (2000, 60000, (64,)), (800, 3000, (3, 3, 3, 3))):
        ei = synthetic.gcl_small(n, e, seed=n).to("cuda")
        for trial in range(3):
            gen = torch.Generator(device="cuda").manual_seed(trial)
            seeds = torch.randperm(n, device="cuda", generator=gen)[: 7 + 40 * trial]
            a, b = NeighborSampler(ei, n, fans, seed=5), NeighborSampler(ei, n, fans, seed=5)
            for _ in range(2):                                # (two calls: the per-call seed advances alike)
                got, want = a.sample(seeds), b.sample_hop_by_hop(seeds)
                for x, y in zip(got, want):
                    assert torch.equal(x, y), (n, fans, trial, x.shape, y.shape)


def test_one_call_saint_sampler_equals_the_torch_glue_sampler():
    from biomedkg_amd import synthetic
    from biomedkg_amd.samplers import SaintRandomWalkSampler
    g = synthetic.primekg_like(4000, 60000, 6, seed=2)
    ei = g["edge_index"].to("cuda")
    for trial in range(3):
        a, b = (SaintRandomWalkSampler(ei, 4000, 48, walk_length=10, seed=9) for _ in range(2))
        start = torch.randint(0, 4000, (48,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(trial))
        for _ in range(2):
            for x, y in zip(a.sample(start), b.sample_with_torch_glue(start)):
                assert torch.equal(x, y)


def test_mini_batch_structures_die_with_their_batch():
    """A structure cached ON a batch's edge_index must not keep the batch alive (tensor -> cache -> structure -> tensor
    was an uncollectable cycle: 2-3 MB left behind per NeighborLoader step)."""
    import gc
    from biomedkg_amd import synthetic
    from biomedkg_amd.gcl_module import GGDModule
    from biomedkg_amd.kge_module import KGEModule
    from biomedkg_amd.samplers import NeighborBatchLoader, SaintBatchLoader
    ei = synthetic.gcl_small(6000, 90000, seed=2).to("cuda")
    x3 = synthetic.lm_features(6000, 2, 32, seed=3).to("cuda")
    gcl = GGDModule(in_dim=32, hidden_dim=32, out_dim=32, num_hidden_layers=1, fuse_method="attention").to("cuda").train()
    g = synthetic.primekg_like(6000, 90000, 5, seed=1)
    kei, ket = g["edge_index"].to("cuda"), g["edge_type"].to("cuda")
    kge = KGEModule("rgcn", "dismult", 32, 32, 32, 1, 5, 2, "cosine", 1e-3, 0.0, "none", 2, "random").to("cuda").train()
    xk = torch.randn(6000, 32, device="cuda")

    def run(steps):
        k = 0
        while k < steps:
            for b in NeighborBatchLoader(x3, ei, None, 128, (30, 30, 30), True, seed=3):
                gcl.training_step(b).backward()
                k += 1
                if k >= steps:
                    break
        for b in SaintBatchLoader(xk, kei, ket, 32, walk_length=10, num_steps=steps, seed=3):
            kge.training_step(b).backward()
        gcl.zero_grad(set_to_none=True); kge.zero_grad(set_to_none=True)
        gc.collect(); torch.cuda.synchronize()
        return torch.cuda.memory_allocated()
    base = run(15)
    grown = run(60) - base
    assert grown < 8 * 2 ** 20, f"{grown / 2 ** 20:.1f} MiB left behind by 60 more steps"


# ------------------------------------------------------------------------------------------- round 4: against the oracle
def _oracle_graphs():
    """(name, edge_index, num_nodes, seeds): a hub + a node without in-edges + a sink + isolated nodes; a graph whose every
    in-degree is below the fan-out; a PrimeKG-like synthetic graph."""
    from biomedkg_amd import synthetic
    g = torch.Generator().manual_seed(11)
    n = 600
    ei = torch.randint(0, n - 20, (2, 24_000), generator=g)           # nodes n-20 .. n-1 stay isolated
    ei[1, :3000] = 7                                                  # hub: ~3,000 in-edges
    ei[0, ei[0] == 13] = 14                                           # 13: a sink (no out-edges)
    ei[1, ei[1] == 21] = 22                                           # 21: no in-edges (a seed that expands to nothing)
    out = [("hub_sink_isolated", ei, n, torch.tensor([7, 21, 13, 590, 3, 250]))]
    ei2 = torch.randint(0, 400, (2, 2400), generator=g)               # mean in-degree 6: fan-out 30 is above every degree
    assert int(torch.bincount(ei2[1], minlength=400).max()) < 30
    out.append(("fanout_above_every_degree", ei2, 400, torch.tensor([1, 2, 3, 399])))
    pk = synthetic.primekg_like(3000, 90_000, 8, seed=3)
    out.append(("primekg_like", pk["edge_index"], pk["num_nodes"], torch.arange(0, 3000, 47)))
    return out


@pytest.mark.parametrize("case", range(3))
@pytest.mark.parametrize("fanouts", [(30, 30, 30), (5, 3), (40, 33)])      # (40, 33: the 64-pick instance of the draw)
def test_neighbor_sampler_equals_the_oracle_on_the_same_tape(case, fanouts):
    """NeighborLoader(num_neighbors=[30] * 3) (reference data_module.py:93-99): the device sampler and oracle/samplers.py
    (pyg-lib's neighbor_sample restated) fed the SAME tape of raw draws give the same batch: seeds first, the same nodes
    discovered in every hop, the same sampled edges (global source, global target, edge id), the relabelled edge_index
    consistent with n_id.  The two order a hop's new nodes differently (discovery order upstream, ascending ids on the
    device): a relabelling of the same batch.  Both device forms (one host read; hop by hop) are held to it."""
    from biomedkg_amd.samplers import NeighborSampler
    from oracle import samplers as S
    name, ei, n, seeds = _oracle_graphs()[case]
    tape = S.make_tape(n, S.TAPE_STRIDE, seed=5 + case)
    want = S.neighbor_sample(ei, n, seeds, list(fanouts), tape)
    w_nodes, w_sub, w_eid = want["n_id"], want["edge_index"], want["e_id"]
    w_triples = sorted(zip(w_nodes[w_sub[0]].tolist(), w_nodes[w_sub[1]].tolist(), w_eid.tolist()))
    samp = NeighborSampler(ei.to(DEV), n, fanouts, seed=123)
    for form in (samp.sample, samp.sample_hop_by_hop):
        n_id, sub, e_id = (t.cpu() for t in form(seeds.to(DEV), tape.to(DEV)))
        assert n_id[: seeds.numel()].tolist() == seeds.tolist(), name
        assert n_id.unique().numel() == n_id.numel()
        assert sorted(n_id.tolist()) == sorted(w_nodes.tolist()), name
        # hop by hop: the device appends a hop's discoveries in ascending id order
        at = 0
        for hop, nodes in enumerate(want["hops"]):
            got = n_id[at:at + len(nodes)].tolist()
            assert got == (nodes if hop == 0 else sorted(nodes)), (name, hop)
            at += len(nodes)
        triples = sorted(zip(n_id[sub[0]].tolist(), n_id[sub[1]].tolist(), e_id.tolist()))
        assert triples == w_triples, name
        assert torch.equal(n_id[sub[0]], ei[0, e_id]) and torch.equal(n_id[sub[1]], ei[1, e_id])
    # without a tape the sampler draws from its own stream: another batch (the hub has 3,000 in-edges to choose from)
    if case == 0:
        _, _, e_free = samp.sample(seeds.to(DEV))
        assert sorted(e_free.cpu().tolist()) != sorted(w_eid.tolist())


@pytest.mark.parametrize("case", range(3))
def test_saint_random_walk_sampler_equals_the_oracle_on_the_same_tape(case):
    """GraphSAINTRandomWalkSampler(walk_length=10) (reference data_module.py:101-107): same roots + same tape -> the same
    walks step for step (sinks and isolated roots stay put), the same node set (ascending), the same induced edges and the
    same relabelled edge_index as oracle/samplers.py (torch-sparse random_walk + saint_subgraph restated), for the one-call
    device batch and the torch-glue form alike."""
    from biomedkg_amd.samplers import SaintRandomWalkSampler
    from oracle import samplers as S
    name, ei, n, seeds = _oracle_graphs()[case]
    g = torch.Generator().manual_seed(case)
    start = torch.cat([seeds, torch.randint(0, n, (64 - seeds.numel(),), generator=g)])[:64]
    tape = S.make_tape(start.numel(), 10, seed=17 + case)
    want = S.saint_random_walk_batch(ei, n, start, 10, tape)
    samp = SaintRandomWalkSampler(ei.to(DEV), n, batch_size=start.numel(), walk_length=10, seed=9)
    walks = samp.walks(start.to(DEV), tape.to(DEV)).cpu()
    assert torch.equal(walks, want["walks"]), name
    for form in (samp.sample, samp.sample_with_torch_glue):
        n_id, sub, e_id = (t.cpu() for t in form(start.to(DEV), tape.to(DEV)))
        assert torch.equal(n_id, want["n_id"]), name
        order = torch.argsort(e_id)
        assert torch.equal(e_id[order], want["e_id"]), name
        assert torch.equal(sub[:, order], want["edge_index"]), name
    with pytest.raises(ValueError):
        samp.sample(None, tape.to(DEV))
