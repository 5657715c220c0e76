"""Node-partitioned GCN / R-GCN (+ DistMult over all-gathered embeddings) on 2 and 3 gloo ranks vs. the CPU
oracle on the whole graph: every shard of the output, and every weight gradient after the cross-rank sum,
must match.  Exercises NodePartition, the padded slot space, the in-edge / out-edge CSRs, the gather-only exchange
(forward tables AND the backward's output gradients travel as gathers; nothing is reduce-scattered) in both of its
transports -- the plain all-gather and the halo exchange that moves only the rows a peer's CSRs reference, whole and
in row chunks.  CPU only (torch arithmetic backend from tests/cpu_backend.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    from biomedkg_amd import synthetic
    g = synthetic.primekg_like(400, 6000, 6, seed=5)
    torch.manual_seed(5)
    return g, torch.randn(400, 24), torch.randn(400, 20)


def _worker(rank, world, port, outdir, halo):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from cpu_backend import TorchBackend as B
    from biomedkg_amd import distributed as D
    from biomedkg_amd import partition as P
    from oracle import ref_modules as R
    D.init_from_env(backend="gloo")
    g, x_gcn, x_rgcn = _problem()
    ei, et, n = g["edge_index"], g["edge_type"], g["num_nodes"]
    part = P.NodePartition(ei, n, rank, world, halo=halo)
    res = {"bounds": part.bounds, "halo_fraction": dict(part.halo_fraction),
           "halo_used": {k: v is not None for k, v in part.comm.patterns.items()}}

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
    res["gcn_wire"] = part.comm.take_stats()[:3]

    # ---- the exchange itself: every row the rank's CSRs read arrives with its owner's content, whole or in row chunks
    comm = part.comm
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
    res["chunks"] = len(chunks)
    comm.take_stats()

    # ---- R-GCN encoder + DistMult over the gathered embeddings, triplets sharded over ranks
    torch.manual_seed(12)
    model = R.make_kge_model("rgcn", "dismult", 20, 16, 16, 1, g["num_relations"]).eval()
    prg = P.PartitionedRelGraph(part, ei, et, g["num_relations"], B)
    x_local = part.shard(x_rgcn).clone().requires_grad_(True)
    z_local = P.rgcn_encoder_partitioned(model.encoder, x_local, prg, B)
    table = P.gather_embeddings(z_local, part)
    mine = torch.arange(rank, ei.size(1), world)                       # this rank's share of the triplets
    h, t, r = part.slots(ei[0, mine]), part.slots(ei[1, mine]), et[mine]
    score = (table[h] * model.decoder.rel_emb[r] * table[t]).sum(1)
    score.sum().backward()
    grads = [p.grad.clone() for p in model.parameters()]
    for gr in grads:
        dist.all_reduce(gr)
    res.update(rgcn_z=z_local.detach(), rgcn_grads=grads, rgcn_dx=x_local.grad.clone(), score_sum=score.sum().detach())

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
    dist.destroy_process_group()


@pytest.mark.parametrize("world,halo", [(2, False), (2, True), (3, True), (3, None)])
def test_partitioned_encoders_match_whole_graph(tmp_path, world, halo):
    from oracle import ref_modules as R
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), halo), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"r{i}.pt")) for i in range(world)]
    g, x_gcn, x_rgcn = _problem()
    ei, et = g["edge_index"], g["edge_type"]
    bounds = parts[0]["bounds"]
    assert bounds[0] == 0 and bounds[-1] == 400 and all(p["bounds"] == bounds for p in parts)
    from biomedkg_amd.partition import NodePartition
    load = NodePartition.load_of(ei, 400, NodePartition.NODE_WEIGHT)
    shares = [float(load[a:b].sum()) for a, b in zip(bounds[:-1], bounds[1:])]
    assert max(shares) < 1.3 * sum(shares) / world                       # balanced by incident edges + the per-node weight
    # what crossed the wire for the 4-layer GCN encoder (4 forward tables + 4 gathered output gradients, fp32 here):
    # the halo exchange moves only the rows the peer's CSRs reference, the all-gather every (padded) row
    n_max = max(b - a for a, b in zip(bounds[:-1], bounds[1:]))
    for rank, p in enumerate(parts):
        received, sent, count = p["gcn_wire"]
        everything = 8 * 16 * 4 * n_max * (world - 1)
        used = p["halo_used"]
        assert halo is None or (used["in"] == used["out"] == bool(halo))
        if used["in"] and used["out"]:
            own = bounds[rank + 1] - bounds[rank]
            assert count == 8 and received <= 8 * 16 * 4 * (400 - own) and received < everything
        elif not used["in"] and not used["out"]:
            assert count == 8 and received == everything
        else:
            assert count == 8 and received < everything
    assert all(0.0 < f <= 1.0 for f in parts[0]["halo_fraction"].values())
    assert all(p["exchange_ok"] for p in parts)
    assert all(p["chunks"] == (4 if p["halo_used"]["in"] else 1) for p in parts)

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

    torch.manual_seed(12)
    model = R.make_kge_model("rgcn", "dismult", 20, 16, 16, 1, g["num_relations"]).eval()
    x = x_rgcn.clone().requires_grad_(True)
    z = model.encode(x, ei, et)
    score = model.decode(z, ei, et)
    score.sum().backward()
    torch.testing.assert_close(torch.cat([p["rgcn_z"] for p in parts]), z.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(sum(p["score_sum"] for p in parts), score.sum().detach(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(torch.cat([p["rgcn_dx"] for p in parts]), x.grad, rtol=1e-3, atol=1e-4)
    for got, p in zip(parts[0]["rgcn_grads"], model.parameters()):
        torch.testing.assert_close(got, p.grad, rtol=1e-3, atol=1e-3)

    # GGD: same permutation and edge mask on the whole graph through the oracle
    from oracle import pygcl_ops as G
    torch.manual_seed(13)
    ggd = R.GGD(R.GCNEncoder(24, 16, 16, 1), 16, 1, 0.5).eval()
    perm = torch.randperm(400, generator=torch.Generator().manual_seed(4))
    pos = ggd.encoder(x_gcn, ei[:, keep])
    neg = ggd.encoder(x_gcn[perm], ei[:, keep])
    head = ggd.mlp[-1]
    pred = torch.cat([head(pos).sum(1), head(neg).sum(1)])
    ref = torch.nn.functional.binary_cross_entropy_with_logits(pred, torch.cat([torch.ones(400), torch.zeros(400)]))
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
