"""oracle/lean.py -- the checkpointed form of the CPU oracle that bench.py times at the full bench size -- IS the oracle:
on the 5,000-node / 50,000-edge graph (BASELINE configs[0] size), with the same recorded draws, both training steps give
the plain oracle's losses (1e-6) and parameter gradients (1e-5, Frobenius).  SURVEY.md 8d: "a checkpointed oracle variant proven
equal on G-small"."""
import copy

import torch

from biomedkg_amd import synthetic
from oracle import lean, rand as orand, ref_modules as R


def _grads(params):
    return [p.grad.clone() for p in params]


def test_lean_steps_equal_the_plain_oracle_on_g_small():
    g = synthetic.primekg_like(5000, 50_000, 6, seed=42)
    ei, et, n = g["edge_index"], g["edge_type"], g["num_nodes"]
    torch.manual_seed(0)
    xg, xk = torch.randn(n, 96), torch.randn(n, 32)
    ggd = R.GGD(R.GCNEncoder(96, 32, 32, 2), 32, 1, 0.5)
    kge = R.make_kge_model("rgcn", "dismult", 32, 32, 32, 2, g["num_relations"])
    ggd2, kge2 = copy.deepcopy(ggd), copy.deepcopy(kge)
    for coin_seed in (0, 1, 2):                                   # both outcomes of GGD's augmentation coin turn up
        torch.manual_seed(100 + coin_seed)
        tape = orand.TapeRand()
        with orand.use(tape):
            want = R.ggd_loss(ggd, xg, ei)
        ggd.zero_grad(); want.backward()
        with orand.use(orand.ReplayRand(tape.tape)):
            got = lean.ggd_loss(ggd2, xg, ei)
        ggd2.zero_grad(); got.backward()
        assert abs(float(got.detach()) - float(want.detach())) <= 1e-6 * abs(float(want.detach()))
        for a, b in zip(_grads(ggd2.parameters()), _grads(ggd.parameters())):
            assert float((a - b).norm() / b.norm().clamp_min(1e-30)) <= 1e-5
    lean_chunk = lean.CHUNK
    lean.CHUNK = 7001                                             # several ragged decoder chunks per triplet set
    try:
        torch.manual_seed(7)
        tape = orand.TapeRand()
        with orand.use(tape):
            want = R.kge_step(kge, xk, ei, et, 2)[0]
        kge.zero_grad(); want.backward()
        with orand.use(orand.ReplayRand(tape.tape)):
            got = lean.kge_step(kge2, xk, ei, et, 2)[0]
        kge2.zero_grad(); got.backward()
    finally:
        lean.CHUNK = lean_chunk
    assert abs(float(got.detach()) - float(want.detach())) <= 1e-6 * abs(float(want.detach()))
    for a, b in zip(_grads(kge2.parameters()), _grads(kge.parameters())):          # (chunked sums: another fp32 order)
        assert float((a - b).norm() / b.norm().clamp_min(1e-30)) <= 1e-5
