"""The node-partitioned layers on the HIP backend (world size 1: the exchange degenerates to a copy) must equal
the ordinary single-GPU layers -- checks HipBackend, the rectangular segment reduce and its transpose, and the
autograd chain through all_gather_rows.  The cross-rank logic is covered on CPU by tests/test_partition_gloo.py.
Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_partitioned_gcn_and_rgcn_world1_match_fused_layers():
    from biomedkg_amd import ops, partition as P, synthetic
    from biomedkg_amd.graph import graph_of, rel_graph_of
    from biomedkg_amd.model.encoder import GCNEncoder, RGCN
    g = synthetic.primekg_like(3000, 60_000, 8, seed=2)
    ei, et, n = g["edge_index"].to(DEV), g["edge_type"].to(DEV), g["num_nodes"]
    torch.manual_seed(0)
    part = P.NodePartition(ei, n, 0, 1)
    assert part.bounds == [0, n] and part.n_max == n

    keep = torch.rand(ei.size(1), device=DEV) >= 0.4
    enc = GCNEncoder(64, 32, 32, 2).to(DEV).eval()
    x = torch.randn(n, 64, device=DEV)
    pg = P.PartitionedGraph(part, ei, P.HipBackend, keep=keep)
    out_p = P.gcn_encoder_partitioned(enc, x, pg, P.HipBackend)
    out_p.pow(2).sum().backward()
    grads_p = [p.grad.clone() for p in enc.parameters()]
    enc.zero_grad()
    out = enc(x, ei, edge_keep=keep)
    out.pow(2).sum().backward()
    torch.testing.assert_close(out_p, out, rtol=1e-4, atol=1e-5)
    for a, p in zip(grads_p, enc.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=1e-3, atol=1e-4)

    renc = RGCN(64, 32, 32, 1, g["num_relations"]).to(DEV).eval()
    prg = P.PartitionedRelGraph(part, ei, et, g["num_relations"], P.HipBackend)
    xr = x.clone().requires_grad_(True)
    z_p = P.rgcn_encoder_partitioned(renc, xr, prg, P.HipBackend)
    z_p.pow(2).sum().backward()
    grads_p, dx_p = [p.grad.clone() for p in renc.parameters()], xr.grad.clone()
    renc.zero_grad()
    xr2 = x.clone().requires_grad_(True)
    z = renc(xr2, ei, et)
    z.pow(2).sum().backward()
    torch.testing.assert_close(z_p, z, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dx_p, xr2.grad, rtol=1e-3, atol=1e-4)
    for a, p in zip(grads_p, renc.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=1e-3, atol=1e-3)


def test_partitioned_training_losses_world1_match_the_modules():
    """``bench.py --mode partition`` at one rank: the partitioned GGD / KGE losses and every parameter gradient equal
    the whole-graph modules' (same corruption permutation, same negatives, dropout off)."""
    from types import SimpleNamespace
    from biomedkg_amd import partition as P, rand as R, sampling, synthetic
    from biomedkg_amd.gcl_module import GGDModule
    from biomedkg_amd.kge_module import KGEModule
    g = synthetic.primekg_like(3000, 60_000, 8, seed=4)
    ei, et, n, r = g["edge_index"].to(DEV), g["edge_type"].to(DEV), g["num_nodes"], g["num_relations"]
    part = P.NodePartition(ei, n, 0, 1)
    torch.manual_seed(0)
    # --- GGD
    gcl = GGDModule(in_dim=64, hidden_dim=32, out_dim=32, num_hidden_layers=2, fuse_method="none").to(DEV).eval()
    x = torch.randn(n, 64, device=DEV)
    perm = torch.randperm(n, device=DEV)
    pg = P.PartitionedGraph(part, ei, P.HipBackend)
    loss_p = P.ggd_loss_partitioned(gcl.model, x, pg, P.HipBackend, perm)
    loss_p.backward()
    grads_p = [p.grad.clone() for p in gcl.model.parameters()]
    gcl.model.zero_grad()
    pos_h = gcl.model.encoder(x, ei)
    neg_h = gcl.model.encoder(x[perm], ei)
    head = gcl.model.mlp[-1]
    pred = torch.cat([torch.nn.functional.linear(pos_h, head.weight, head.bias).sum(1),
                      torch.nn.functional.linear(neg_h, head.weight, head.bias).sum(1)])
    gt = torch.cat([torch.ones(n, device=DEV), torch.zeros(n, device=DEV)])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, gt)
    loss.backward()
    torch.testing.assert_close(loss_p, loss, rtol=1e-4, atol=1e-5)
    for a, p in zip(grads_p, gcl.model.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=2e-3, atol=1e-4)
    # --- KGE
    kge = KGEModule("rgcn", "dismult", 64, 32, 32, 2, r, 1, "cosine", 1e-3, 0.1, "none", 1, "random").to(DEV).eval()
    neg = sampling.negative_sampling(ei, n, ei.size(1), seed=3)
    nt = et[torch.randperm(et.numel(), device=DEV)]
    prg = P.PartitionedRelGraph(part, ei, et, r, P.HipBackend)
    loss_p = P.kge_loss_partitioned(kge.model, x, prg, P.HipBackend, ei, et, neg, nt, kge.model.decoder.kind)
    loss_p.backward()
    grads_p = [p.grad.clone() for p in kge.model.parameters()]
    kge.model.zero_grad()
    z = kge.model.encode(x, ei, et)
    pred = torch.cat([kge.model.decode(z, ei, et), kge.model.decode(z, neg, nt)])
    gt = torch.cat([torch.ones(ei.size(1), device=DEV), torch.zeros(neg.size(1), device=DEV)])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, gt) + 1e-2 * (
        z.pow(2).mean() + kge.model.decoder.rel_emb.pow(2).mean())
    loss.backward()
    torch.testing.assert_close(loss_p, loss, rtol=1e-4, atol=1e-5)
    for a, p in zip(grads_p, kge.model.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=2e-3, atol=1e-4)
