#!/usr/bin/env python3
"""The reference's GCL regime (train_gcl.py:110: NeighborLoader batches, 128 seeds, fan-out [30]*3) on a single-type
graph: sampler vs step time of the eager loop.  One JSON line."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biomedkg_amd import synthetic
from biomedkg_amd.gcl_module import GGDModule, GRACEModule
from biomedkg_amd.samplers import NeighborBatchLoader

def main(n=27_000, e=640_000, seeds=128, steps=30, model="ggd"):
    dev = torch.device("cuda:0")
    ei = synthetic.gcl_small(n, e, seed=1).to(dev)
    x = synthetic.lm_features(n, 2, 768, seed=2).to(dev)
    torch.manual_seed(0)
    cls = GGDModule if model == "ggd" else GRACEModule
    m = cls(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="attention").to(dev).train()
    params = list(m.model.parameters())
    opt = torch.optim.Adam(params, lr=1e-3, fused=True)
    def run(k):
        it = iter(NeighborBatchLoader(x, ei, None, seeds, (30, 30, 30), True, seed=3))
        ts = tt = 0.0; nn = ne = 0
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
            ts += t1 - t0; tt += t2 - t1; nn += b.x.size(0); ne += b.edge_index.size(1)
        return ts / k, tt / k, nn / k, ne / k
    run(5)
    ts, tt, nn, ne = run(steps)
    print(json.dumps({"model": model, "graph": [n, e], "mean_nodes": nn, "mean_edges": ne, "sampler_ms": round(ts * 1e3, 2),
                      "step_ms": round(tt * 1e3, 2)}))

if __name__ == "__main__":
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(**{k: (v if k == "model" else int(v)) for k, v in kw.items()})
