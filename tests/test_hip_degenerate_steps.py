"""Degenerate-size training steps (graphs of 2..60 nodes, 0..200 edges, more negatives asked for than free cells,
self loops, duplicates) through the eager KGE / GGD / GRACE / DGI steps in both precision modes: a finite loss, NaN
only on an edgeless KGE batch (the mean over no triplets, as in the reference), or a Python exception where the
reference fails too (fewer negatives than neg_ratio * E) -- never an out-of-bounds access (scripts/fuzz_tiny.py is the
long form of this sweep).  Needs an MI355X."""
import random
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_tiny_and_saturated_graphs_never_fault():
    from biomedkg_amd import ops, rand as R
    from biomedkg_amd.gcl_module import DGIModule, GGDModule, GRACEModule
    from biomedkg_amd.kge_module import KGEModule
    rnd = random.Random(11)
    raised = finite = 0
    for it in range(40):
        n, e = rnd.choice([2, 3, 5, 10, 60]), rnd.choice([0, 1, 2, 5, 20, 200])
        r, ratio, d = rnd.randint(1, 4), rnd.choice([0, 1, 3, 50]), rnd.choice([16, 64])
        ops.set_matmul_precision(rnd.choice(["fp32", "bf16"]))
        g = torch.Generator().manual_seed(it)
        ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
        et = torch.randint(0, r, (e,), generator=g).to(DEV)
        what = rnd.choice(["kge-dismult", "kge-transe", "ggd", "grace", "dgi"])
        torch.manual_seed(it)
        if what.startswith("kge"):
            m = KGEModule("rgcn", what[4:], d, d, d, 1, r, 2, "cosine", 1e-3, 0.0, "none", ratio, "random").to(DEV).train()
            batch = SimpleNamespace(x=torch.randn(n, d, generator=g).to(DEV), edge_index=ei, edge_type=et)
        else:
            cls = {"ggd": GGDModule, "grace": GRACEModule, "dgi": DGIModule}[what]
            m = cls(in_dim=d, hidden_dim=d, out_dim=d, num_hidden_layers=1, fuse_method="attention").to(DEV).train()
            batch = SimpleNamespace(x=torch.randn(n, 2, d, generator=g).to(DEV), edge_index=ei)
        try:
            with R.use(R.DeviceRand(seed=it)):
                loss = m.training_step(batch)
                loss.backward()
            torch.cuda.synchronize()
        except ValueError as why:                       # the reference's broadcast error, refused up front here
            assert "do not match" in str(why), why
            raised += 1
            continue
        val = float(loss)
        if e == 0 and what.startswith("kge"):
            continue
        assert val == val and abs(val) != float("inf"), (it, what, n, e, r, ratio, val)
        assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in m.model.parameters()), (it, what)
        finite += 1
    assert finite >= 20 and raised >= 1
