"""tests/bf16_model.py (the fp64 restatement of the bf16 path that the full-size GPU test compares with) is itself held
to the oracle here: with its bf16 rounding switched off it must BE the oracle's GGD step and R-GCN / DistMult step --
same loss, same gradients, to float64 round-off -- so whatever the GPU test finds is about rounding points and
accumulation, not about a second, unpinned formulation of the model."""
import torch

from oracle import rand as orand
from oracle import ref_modules as O
from tests import bf16_model as B


class _NoRounding:
    def __enter__(self):
        self.saved = (B.rf, B.rb, B.bf)
        B.rf = B.rb = lambda t: t
        B.bf = lambda t: t.double()

    def __exit__(self, *exc):
        B.rf, B.rb, B.bf = self.saved


class _Fixed(orand.TorchRand):
    def __init__(self, perm):
        self._perm = perm

    def coin(self):
        return torch.tensor(1.0)          # >= p: no augmentation

    def perm(self, n):
        return self._perm


def test_ggd_restatement_without_rounding_is_the_oracle():
    torch.manual_seed(0)
    n, e, d_in, h = 70, 500, 24, 16
    ei = torch.randint(0, n, (2, e))
    ei[:, :3] = torch.tensor([[4, 4, 9], [4, 4, 9]])                 # (duplicate) self loops: add_remaining_self_loops
    x = torch.randn(n, d_in, dtype=torch.float64)
    model = O.GGD(O.GCNEncoder(d_in, h, h, 2, drop_out=False), h, 1, 0.5).double()
    for layer in model.encoder.graph_layers:
        with torch.no_grad():
            layer.bias.uniform_(-0.2, 0.2)
    perm = torch.randperm(n)
    with orand.use(_Fixed(perm)):
        want = O.ggd_loss(model, x, ei)
    want.backward()
    cw = [B.leaf(l.lin.weight) for l in model.encoder.graph_layers]
    cb = [B.leaf(l.bias) for l in model.encoder.graph_layers]
    hw, hb = B.leaf(model.mlp[0].weight), B.leaf(model.mlp[0].bias)
    with _NoRounding():
        got = B.ggd_step(x, ei, perm, cw, cb, hw, hb)
        got.backward()
    assert abs(float(got) - float(want)) < 1e-12
    for mine, ref in zip(cw + cb + [hw, hb], [l.lin.weight for l in model.encoder.graph_layers] +
                         [l.bias for l in model.encoder.graph_layers] + [model.mlp[0].weight, model.mlp[0].bias]):
        torch.testing.assert_close(mine.grad, ref.grad, rtol=1e-9, atol=1e-12)


def test_kge_restatement_without_rounding_is_the_oracle():
    torch.manual_seed(1)
    n, e, r, d = 60, 400, 4, 16
    ei, et = torch.randint(0, n, (2, e)), torch.randint(0, r, (e,))
    et[et == 2] = 3                                                  # an empty relation
    x = torch.randn(n, d, dtype=torch.float64)
    model = O.make_kge_model("rgcn", "dismult", d, d, d, 1, r).double()
    model.encoder.drop_out = False
    for layer in model.encoder.graph_layers:
        with torch.no_grad():
            layer.bias.uniform_(-0.2, 0.2)
    tape = orand.TapeRand()
    with orand.use(tape):
        want, z, _, _ = O.kge_step(model, x, ei, et, 2)
    want.backward()
    draws = dict((k, v) for k, v in tape.tape)
    neg, nperm = draws["neg_edges"], draws["perm"]
    nt = et.repeat(2)[nperm]
    layers = model.encoder.graph_layers
    W, Rt, b = [B.leaf(l.weight) for l in layers], [B.leaf(l.root) for l in layers], [B.leaf(l.bias) for l in layers]
    rel = B.leaf(model.decoder.rel_emb)
    with _NoRounding():
        got, z2 = B.kge_step(x, ei, et, neg, nt, W, Rt, b, rel, r)
        got.backward()
    assert abs(float(got) - float(want)) < 1e-12
    torch.testing.assert_close(z2, z, rtol=1e-9, atol=1e-12)
    for mine, ref in zip(W + Rt + b + [rel], [l.weight for l in layers] + [l.root for l in layers] + [l.bias for l in layers]
                         + [model.decoder.rel_emb]):
        torch.testing.assert_close(mine.grad, ref.grad, rtol=1e-9, atol=1e-12)


def test_rounding_hooks_round_values_and_gradients():
    x = torch.tensor([1.0 + 2.0 ** -10, -3.0 - 2.0 ** -9], dtype=torch.float64, requires_grad=True)
    y = B.rf(x)
    assert y.tolist() == [1.0, -3.0]                                # bf16 keeps 8 significant bits
    (y * torch.tensor([1.0 + 2.0 ** -10, 2.0], dtype=torch.float64)).sum().backward()
    assert x.grad.tolist() == [1.0 + 2.0 ** -10, 2.0]               # straight-through
    x.grad = None
    (B.rb(x) * torch.tensor([1.0 + 2.0 ** -10, 2.0], dtype=torch.float64)).sum().backward()
    assert x.grad.tolist() == [1.0, 2.0]                            # the gradient is what gets rounded
