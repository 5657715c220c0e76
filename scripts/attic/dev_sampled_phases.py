#!/usr/bin/env python3
"""Where does a replayed mini-batch step spend its time?  Captures prefixes of the padded step (forward only, forward +
backward, the whole step) for one batch and times their replays."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biomedkg_amd import synthetic, rand as R, _lib as L
from biomedkg_amd.kge_module import KGEModule
from biomedkg_amd.sampled import SaintGraphedKge

dev = torch.device("cuda:0")
g = synthetic.primekg_like(130_000, 4_000_000, 30, seed=1)
ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
x = torch.randn(130_000, 256, device=dev) * 0.05
torch.manual_seed(0)
m = KGEModule("rgcn", "dismult", 256, 256, 256, 2, 30, 2, "cosine", 1e-3, 0.0, "none", 10, "random").to(dev).train()
sg = SaintGraphedKge(m, x, ei, et, 128, 10, None, clip=1.0, seed=3, graphs=False)
for _ in range(3):
    sg.step()
n_b, e_b, e_free = sg.last_sizes
bucket = sg.bucket_of(n_b, e_b, e_free)
out = {"batch": sg.last_sizes, "bucket": bucket}
L.call("bkg_set_rng_epoch", L.ptr(sg.epoch))

def variant(name):
    def body():
        sg.rand.begin_step()
        if name == "sampler":
            sg.sampler._draw(); return
        sg.optimizer.zero_grad(set_to_none=True)
        loss = sg.padded_loss(bucket)
        if name in ("fwd_bwd", "step"):
            loss.backward()
        if name == "step":
            torch.nn.utils.clip_grad_norm_(sg.params, 1.0); sg.optimizer.step()
    return body

with R.use(sg.rand), torch.cuda.stream(sg.stream):
    for name in ("sampler", "fwd", "fwd_serial", "fwd_bwd", "step"):
        sg.branches = name != "fwd_serial"
        body = variant("fwd" if name == "fwd_serial" else name)
        body(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            body()
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100):
            gr.replay()
        torch.cuda.synchronize()
        out[name + "_us"] = round((time.perf_counter() - t0) / 100 * 1e6, 1)
print(json.dumps(out))
