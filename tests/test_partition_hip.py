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
