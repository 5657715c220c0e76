#!/usr/bin/env python3
"""The reference's default KGE regime (train_kge.py:86: GraphSAINT random-walk batches, batch_size 128, walk_length 10,
neg_ratio 10, R-GCN 256-256-256 + DistMult) on the synthetic PrimeKG-shaped graph: wall time per step split into the
sampler and the training step, eager.  Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biomedkg_amd import ops, synthetic                      # noqa: E402
from biomedkg_amd.kge_module import KGEModule                # noqa: E402
from biomedkg_amd.samplers import SaintBatchLoader           # noqa: E402


def main(n=130_000, e=4_000_000, walks=128, steps=100, dim=256, neg_ratio=10, only_graphed=0):
    dev = torch.device("cuda:0")
    g = synthetic.primekg_like(n, e, 30, seed=1)
    ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
    x = torch.randn(n, dim, device=dev) * 0.05
    torch.manual_seed(0)
    m = KGEModule(encoder_name="rgcn", decoder_name="dismult", in_dim=dim, hidden_dim=dim, out_dim=dim,
                  num_hidden_layers=2, num_relation=g["num_relations"], num_heads=2, scheduler_type="cosine",
                  learning_rate=1e-3, warm_up_ratio=0.0, fuse_method="none", neg_ratio=neg_ratio,
                  node_init_method="random").to(dev)
    opt = torch.optim.Adam(m.model.parameters(), lr=1e-3, fused=True)
    loader = SaintBatchLoader(x, ei, et, walks, walk_length=10, num_steps=steps, seed=3)
    params = list(m.model.parameters())

    def run(k):
        t_s = t_t = 0.0
        nn = ne = 0
        it = iter(SaintBatchLoader(x, ei, et, walks, walk_length=10, num_steps=k, seed=3))
        for _ in range(k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            b = next(it)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            loss = m.training_step(b)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            t_s += t1 - t0; t_t += t2 - t1
            nn += b.x.size(0); ne += b.edge_index.size(1)
        return t_s / k, t_t / k, nn / k, ne / k

    out = {"regime": "saint-rw", "walks": walks, "neg_ratio": neg_ratio}
    if not only_graphed:
        run(10)
        ts, tt, nn, ne = run(steps)
        out.update({"mean_nodes": nn, "mean_edges": ne, "eager": {"sampler_ms": round(ts * 1e3, 3),
                    "step_ms": round(tt * 1e3, 3), "steps_per_s": round(1.0 / (ts + tt), 1)}})
    # the same regime through the padded, sync-free loop: eagerly, then as one hipGraph replay per step
    from biomedkg_amd.sampled import SaintGraphedKge
    for name, graphs in ((("padded_eager", False),) if not only_graphed else ()) + (("graphed", True), ("graphed_deterministic", True)):
        ops.set_deterministic(name.endswith("deterministic"))
        opt2 = torch.optim.Adam(m.model.parameters(), lr=1e-3, fused=True)
        sg = SaintGraphedKge(m, x, ei, et, walks, 10, opt2, clip=1.0, seed=3, graphs=graphs)
        for _ in range(60 if graphs else 10):
            sg.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        before = dict(sg.stats)
        for _ in range(steps):
            sg.step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        out[name] = {"step_ms": round(dt * 1e3, 3), "steps_per_s": round(1.0 / dt, 1),
                     "stats": {k: sg.stats[k] - before[k] for k in sg.stats}}
    ops.set_deterministic(False)
    print(json.dumps(out))


if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(**{k: int(v) for k, v in kw.items()})
