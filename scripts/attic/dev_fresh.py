#!/usr/bin/env python3
"""Time the structure builds a NEW edge_index costs at bench scale (each on its own, device-synchronised)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biomedkg_amd import synthetic, graph as G, ops, sampling

dev = torch.device("cuda:0")
g = synthetic.primekg_like(130_000, 4_000_000, 30, seed=42)
ei0, et0 = g["edge_index"].to(dev), g["edge_type"].to(dev)
n, r = 130_000, 30
out = {}

def timed(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    out[name] = round((time.perf_counter() - t0) / reps * 1e3, 2)

timed("clone", lambda: ei0.clone())
timed("Graph (2 CSRs)", lambda: G.Graph(ei0.clone(), n))
timed("gcn_weights", lambda: G.Graph(ei0.clone(), n).gcn_weights())
timed("RelGraph", lambda: G.RelGraph(ei0.clone(), et0.clone(), n, r))
timed("Incidence", lambda: ops.Incidence(ei0.clone(), n))
timed("relation_order", lambda: ops.relation_order_of(et0.clone(), r))
timed("positive hash table", lambda: sampling._positive_table(ei0.clone(), n))
timed("loop_free + max", lambda: (sampling._loop_free_count(ei0.clone()), sampling._num_nodes_of(ei0.clone())))
timed("negative_sampling (new graph)", lambda: sampling.negative_sampling(ei0.clone(), n, 4_000_000, seed=5))
x = torch.randn(n, 768, device=dev)
timed("bf16 feature copy", lambda: ops.to_bf16(x))
print(json.dumps(out))
