"""Node-partitioned GCN / R-GCN / KGE / GGD / NT-Xent on 2, 3, 4 and 8 gloo ranks vs. the CPU oracle on the whole graph:
every shard of the output, every weight gradient after the cross-rank sum and every loss must match.  The graph's
load-balanced shards are as skewed as the bench graph's (world 8: 17..84 of 400 rows -- a shard below 5 % of N, row
chunks with empty per-peer lists).  Exercises NodePartition, the padded slot space, the in-edge / out-edge CSRs, the
gather-only exchange (forward tables AND the backward's output gradients travel as gathers; nothing is reduce-scattered)
in its transports -- the plain all-gather, the exact every-row pattern for unequal shards, the halo exchange that moves
only the rows a peer's CSRs reference, whole and in row chunks --, both-owner scoring of the KGE triplets (DistMult and
TransE; no embedding gradient crosses the wire), the first R-GCN layer's constant input gathered once, GGD's paired first
product and the row-partitioned NT-Xent.  CPU only (torch arithmetic backend from tests/cpu_backend.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

N = 400


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    from biomedkg_amd import synthetic
    g = synthetic.primekg_like(N, 6000, 6, seed=5)
    torch.manual_seed(5)
    return g, torch.randn(N, 24), torch.randn(N, 20)


def _negatives(g, k):
    """k distinct non-edge pairs + relation ids, the same on every rank (what the block sampler guarantees on the GPU)."""
    ei, et = g["edge_index"], g["edge_type"]
    gen = torch.Generator().manual_seed(21)
    pos = set((ei[0] * N + ei[1]).tolist())
    cand = torch.randperm(N * N, generator=gen).tolist()
    keep = [c for c in cand if c not in pos and c // N != c % N][:k]
    neg = torch.tensor([[c // N for c in keep], [c % N for c in keep]])
    nt = et.repeat((k + et.numel() - 1) // et.numel())[torch.randperm(k, generator=gen)]
    return neg, nt


def _worker(rank, world, port, outdir, halo):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import torch.distributed as dist
    from cpu_backend import TorchBackend as B
    from biomedkg_amd import distributed as D
    from biomedkg_amd import partition as P
    from oracle import ref_modules as R
    D.init_from_env(backend="gloo")
    g, x_gcn, x_rgcn = _problem()
    ei, et, n = g["edge_index"], g["edge_type"], g["num_nodes"]
    part = P.NodePartition(ei, n, rank, world, halo=halo)
    comm = part.comm
    res = {"bounds": part.bounds, "halo_fraction": dict(part.halo_fraction), "padding_waste": part.padding_waste,
           "halo_used": {k: (v is not None and v is not comm.patterns.get("all")) for k, v in comm.patterns.items()
                         if k != "all"},
           "exact_all": comm.patterns.get("all") is not None}
    pat = comm.patterns.get("in")
    res["empty_peer_lists"] = (sum(1 for prt in pat.parts for q, c in enumerate(prt[1]) if q != rank and c == 0)
                               if pat is not None else 0)

    # ---- GCN encoder (eval mode: ReLU between layers, no dropout)
    torch.manual_seed(11)
    enc = R.GCNEncoder(24, 16, 16, 2).eval()
    for l in enc.graph_layers:
        with torch.no_grad():
            l.bias.uniform_(-0.1, 0.1)
    keep = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(3)) >= 0.3     # a dropout_edge view
    pg = P.PartitionedGraph(part, ei, B, keep=keep)
    out = P.gcn_encoder_partitioned(enc, part.shard(x_gcn), pg, B)
    out.pow(2).sum().backward()
    grads = [p.grad.clone() for p in enc.parameters()]
    for t in grads:
        dist.all_reduce(t)
    res["gcn_out"], res["gcn_grads"] = out.detach(), grads
    res["gcn_wire"] = comm.take_stats()[:3]

    # ---- the exchange itself: every row the rank's CSRs read arrives with its owner's content, whole or in row chunks
    shard = part.shard(x_gcn).contiguous()
    whole = comm.exchange(shard.clone(), "in")
    buf = comm.shard_buffer(shard.size(1), shard.dtype, shard.device)
    chunks = comm.chunks("in")
    pend = []
    for c, (lo, hi) in enumerate(chunks):
        buf[lo:hi] = shard[lo:hi]                                       # "computed" chunk by chunk
        pend.append(comm.exchange_begin(buf, "in", chunk=c if len(chunks) > 1 else None))
    for pnd in pend:
        chunked = comm.exchange_end(pnd)
    slot_node = torch.full((part.world * part.n_max,), -1, dtype=torch.int64)
    slot_node[part.node_slots()] = torch.arange(n)
    read = torch.unique(pg.by_dst.col)
    res["exchange_ok"] = bool(torch.equal(whole[read], x_gcn[slot_node[read]]) and torch.equal(chunked[read], whole[read]))
    back = torch.unique(pg.by_src.col)                                  # pattern "out": the rows the out-edges point at
    res["exchange_ok"] &= bool(torch.equal(comm.exchange(shard.clone(), "out")[back], x_gcn[slot_node[back]]))
    everything = comm.all_gather(shard.clone())                         # pattern "all": every real row, wherever it lives
    res["exchange_ok"] &= bool(torch.equal(everything[part.node_slots()], x_gcn))
    res["chunks"] = len(chunks)
    res["all_wire"] = None
    comm.take_stats()
    comm.all_gather(shard.clone())
    res["all_wire"] = comm.take_stats()[0]

    # ---- R-GCN encoder + the KGE loss with both-owner scoring (DistMult, then TransE): every rank takes, from the SAME
    # global triplet lists, the triplets that touch its nodes; nothing comes back over the wire for dz
    neg, nt = _negatives(g, 2 * ei.size(1))
    pos_share, neg_share = P.TripletShare(part, ei, et), P.TripletShare(part, neg, nt)
    res["own_counts"] = (pos_share.n_own, neg_share.n_own, pos_share.index.size(1), neg_share.index.size(1))
    prg = P.PartitionedRelGraph(part, ei, et, g["num_relations"], B)
    for dec in ("dismult", "transe"):
        torch.manual_seed(12)
        model = R.make_kge_model("rgcn", dec, 20, 16, 16, 1, g["num_relations"]).eval()
        x_local = part.shard(x_rgcn).clone().requires_grad_(True)
        comm.take_stats()
        loss = P.kge_loss_partitioned(model, x_local, prg, B, pos_share, neg_share, 0 if dec == "dismult" else 1,
                                      n_pos_global=ei.size(1), n_neg_global=neg.size(1))
        loss.backward()
        wire = comm.take_stats()[:3]
        grads = [p.grad.clone() for p in model.parameters()]
        for gr in grads:
            dist.all_reduce(gr)
            gr /= world
        res[dec] = dict(loss=P.loss_value(loss, world), grads=grads, dx=x_local.grad.clone() / world, wire=wire)
    # ... and with CONSTANT input features (no gradient asked): the first layer's table is gathered once, not per step
    x_const = part.shard(x_rgcn).clone()
    wires = []
    for _ in range(2):
        model.zero_grad()
        comm.take_stats()
        loss = P.kge_loss_partitioned(model, x_const, prg, B, pos_share, neg_share, 1,
                                      n_pos_global=ei.size(1), n_neg_global=neg.size(1))
        loss.backward()
        wires.append(comm.take_stats()[:3])
    res["const_wires"], res["const_loss"] = wires, P.loss_value(loss, world)
    res["conv_layers"] = len(model.encoder.graph_layers)

    # ---- whole GGD loss (corruption permutation + dropout_edge mask shared by the ranks) and row-partitioned NT-Xent
    from cpu_backend import TorchNtXentKernels
    torch.manual_seed(13)
    ggd = R.GGD(R.GCNEncoder(24, 16, 16, 1), 16, 1, 0.5).eval()
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(4))
    pg2 = P.PartitionedGraph(part, ei, B)
    loss = P.ggd_loss_partitioned(ggd, part.shard(x_gcn), pg2, B, perm, keep=keep)
    loss.backward()
    grads = [p.grad.clone() for p in ggd.parameters()]
    for gr in grads:
        dist.all_reduce(gr)
        gr /= world                                                    # what the gradient bucket does
    res.update(ggd_loss=P.loss_value(loss, world), ggd_grads=grads)
    h1 = part.shard(x_gcn)[:, :12].clone().requires_grad_(True)
    h2 = part.shard(x_gcn)[:, 12:].clone().requires_grad_(True)
    nl = P.ntxent_partitioned(h1, h2, 0.2, part, TorchNtXentKernels())
    nl.backward()
    res.update(ntx_loss=P.loss_value(nl, world), ntx_g1=h1.grad / world, ntx_g2=h2.grad / world)
    torch.save(res, os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,halo", [(2, False), (2, True), (3, True), (3, None), (4, True), (4, None), (8, True),
                                        (8, False), (8, None)])
def test_partitioned_steps_match_whole_graph(tmp_path, world, halo):
    from oracle import ref_modules as R
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), halo), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"r{i}.pt")) for i in range(world)]
    g, x_gcn, x_rgcn = _problem()
    ei, et = g["edge_index"], g["edge_type"]
    bounds = parts[0]["bounds"]
    assert bounds[0] == 0 and bounds[-1] == N and all(p["bounds"] == bounds for p in parts)
    sizes = [b - a for a, b in zip(bounds[:-1], bounds[1:])]
    from biomedkg_amd.partition import NodePartition
    load = NodePartition.load_of(ei, N, NodePartition.NODE_WEIGHT)
    shares = [float(load[a:b].sum()) for a, b in zip(bounds[:-1], bounds[1:])]
    assert max(shares) < 1.3 * sum(shares) / world                       # balanced by incident edges + the per-node weight
    if world == 8:
        assert min(sizes) < 0.05 * N                                     # the skew of the bench graph's shards: a rank
        assert max(sizes) > 4 * min(sizes)                               # with < 5 % of the rows next to one with 20 %
        if halo:
            assert sum(p["empty_peer_lists"] for p in parts) > 0         # row chunks with nothing for some peer
    # what crossed the wire for the 4-layer GCN encoder (4 forward tables + 4 gathered output gradients, fp32 here):
    # the halo exchange moves only the rows the peer's CSRs reference, the all-gather every (padded) row, the exact
    # every-row pattern (unequal shards) the rows the other ranks really hold
    n_max = max(sizes)
    exact_all = parts[0]["exact_all"]
    assert all(p["exact_all"] == exact_all for p in parts)
    assert exact_all == (halo is not False and parts[0]["padding_waste"] > 1.1)
    for rank, p in enumerate(parts):
        received, sent, count = p["gcn_wire"]
        own = sizes[rank]
        padded, real = 8 * 16 * 4 * n_max * (world - 1), 8 * 16 * 4 * (N - own)
        used = p["halo_used"]
        assert halo is None or (used["in"] == used["out"] == bool(halo))
        if used["in"] and used["out"]:
            assert count == 8 and received <= real and received < padded
        elif not used["in"] and not used["out"]:
            assert count == 8 and received == (real if exact_all else padded)
        else:
            assert count == 8 and received < padded
        assert p["all_wire"] == 24 * 4 * ((N - own) if exact_all else n_max * (world - 1))
    assert all(0.0 < f <= 1.0 for f in parts[0]["halo_fraction"].values())
    assert all(p["exchange_ok"] for p in parts)
    assert all(p["chunks"] == (4 if (p["halo_used"]["in"] or exact_all) else 1) for p in parts)

    torch.manual_seed(11)
    enc = R.GCNEncoder(24, 16, 16, 2).eval()
    for l in enc.graph_layers:
        with torch.no_grad():
            l.bias.uniform_(-0.1, 0.1)
    keep = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(3)) >= 0.3
    ref = enc(x_gcn, ei[:, keep])
    ref.pow(2).sum().backward()
    torch.testing.assert_close(torch.cat([p["gcn_out"] for p in parts]), ref.detach(), rtol=1e-4, atol=1e-5)
    for got, p in zip(parts[0]["gcn_grads"], enc.parameters()):
        torch.testing.assert_close(got, p.grad, rtol=1e-4, atol=1e-5)

    # KGE: R-GCN + scorer + BCE + L2 on the whole graph through the oracle, same negatives
    neg, nt = _negatives(g, 2 * ei.size(1))
    assert sum(p["own_counts"][0] for p in parts) == ei.size(1)          # every triplet is counted by exactly one rank
    assert sum(p["own_counts"][1] for p in parts) == neg.size(1)
    if world > 1:                                                        # ... and scored by the owners of both endpoints
        assert sum(p["own_counts"][2] for p in parts) > ei.size(1)
    for dec in ("dismult", "transe"):
        torch.manual_seed(12)
        model = R.make_kge_model("rgcn", dec, 20, 16, 16, 1, g["num_relations"]).eval()
        x = x_rgcn.clone().requires_grad_(True)
        z = model.encode(x, ei, et)
        pred = torch.cat([model.decode(z, ei, et), model.decode(z, neg, nt)])
        gt = torch.cat([torch.ones(ei.size(1)), torch.zeros(neg.size(1))])
        ref = torch.nn.functional.binary_cross_entropy_with_logits(pred, gt) + 1e-2 * (
            z.pow(2).mean() + model.decoder.rel_emb.pow(2).mean())
        ref.backward()
        for p in parts:
            torch.testing.assert_close(p[dec]["loss"], ref.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(torch.cat([p[dec]["dx"] for p in parts]), x.grad, rtol=1e-3, atol=1e-6)
        for got, prm in zip(parts[0][dec]["grads"], model.parameters()):
            torch.testing.assert_close(got, prm.grad, rtol=1e-3, atol=1e-6)
        # the wire of that step: L forward input tables + L gathered dOut + the scorer's table; NOTHING for dz
        for p in parts:
            assert p[dec]["wire"][2] == 2 * p["conv_layers"] + 1
    for p in parts:      # constant features: no dOut gather for the first layer, and from the second step on no input gather
        (_, _, c1), (_, _, c2) = p["const_wires"]
        layers = p["conv_layers"]
        assert (c1, c2) == (2 * layers, 2 * layers - 1)
        torch.testing.assert_close(p["const_loss"], ref.detach(), rtol=1e-5, atol=1e-6)

    # GGD: same permutation and edge mask on the whole graph through the oracle
    from oracle import pygcl_ops as G
    torch.manual_seed(13)
    ggd = R.GGD(R.GCNEncoder(24, 16, 16, 1), 16, 1, 0.5).eval()
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(4))
    pos = ggd.encoder(x_gcn, ei[:, keep])
    neg_h = ggd.encoder(x_gcn[perm], ei[:, keep])
    head = ggd.mlp[-1]
    pred = torch.cat([head(pos).sum(1), head(neg_h).sum(1)])
    ref = torch.nn.functional.binary_cross_entropy_with_logits(pred, torch.cat([torch.ones(N), torch.zeros(N)]))
    ref.backward()
    for p in parts:
        torch.testing.assert_close(p["ggd_loss"], ref.detach(), rtol=1e-5, atol=1e-6)
    for got, p in zip(parts[0]["ggd_grads"], ggd.parameters()):
        torch.testing.assert_close(got, p.grad, rtol=1e-4, atol=1e-6)
    # NT-Xent: the 2N rows partitioned like the nodes vs the oracle on all of them
    h1, h2 = x_gcn[:, :12].clone().requires_grad_(True), x_gcn[:, 12:].clone().requires_grad_(True)
    ref = G.dual_branch_l2l_intraview(h1, h2, 0.2)
    ref.backward()
    for p in parts:
        torch.testing.assert_close(p["ntx_loss"], ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(torch.cat([p["ntx_g1"] for p in parts]), h1.grad, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(torch.cat([p["ntx_g2"] for p in parts]), h2.grad, rtol=1e-4, atol=1e-7)
