"""Dev: where does the bf16 KGE step differ from tests/bf16_model.py?  Compares z, scores, dz and every gradient."""
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops, rand as R, synthetic  # noqa: E402
from biomedkg_amd.kge_module import KGEModule  # noqa: E402
from tests import bf16_model as B  # noqa: E402

DEV = "cuda"
n, r, h = 130_000, 30, 256
g = synthetic.primekg_like(n, 4_000_000, r, seed=42)
ei, et = g["edge_index"].to(DEV), g["edge_type"].to(DEV)
torch.manual_seed(42)
x = torch.randn(n, h, device=DEV)
torch.manual_seed(1234)
kge = KGEModule("rgcn", "dismult", h, h, h, 2, r, 2, "cosine", 1e-3, 0.2, "none", 1, "random").to(DEV).train()
kge.model.encoder.drop_out = False
with torch.no_grad():
    for layer in kge.model.encoder.graph_layers:
        layer.bias.uniform_(-0.05, 0.05)
ops.set_matmul_precision("bf16")
from biomedkg_amd import sampling  # noqa: E402
neg = sampling.negative_sampling(ei, n, ei.size(1), seed=5)
nt = et[torch.randperm(et.numel(), device=DEV)]
z = kge.model.encode(x, ei, et)
z.retain_grad()
dec = kge.model.decoder
loss, pp, npd = ops.kge_bce_loss(z, dec.rel_emb, ei, et, neg, nt, dec.kind, reg=1e-2)
loss.backward()
ops.set_matmul_precision("fp32")

layers = kge.model.encoder.graph_layers
W = [B.leaf(l.weight) for l in layers]; Rt = [B.leaf(l.root) for l in layers]; bs = [B.leaf(l.bias) for l in layers]
rel = B.leaf(dec.rel_emb)
loss64, z64 = B.kge_step(x.double(), ei, et, neg, nt, W, Rt, bs, rel, r)
z64.retain_grad()
loss64.backward()


def err(a, b):
    return float((a.double() - b).norm() / b.norm().clamp_min(1e-30))


print("loss", float(loss), float(loss64))
print("z", err(z.detach(), z64.detach()))
print("dz", err(z.grad, z64.grad))
for i, l in enumerate(layers):
    print(i, "W", err(l.weight.grad, W[i].grad), "root", err(l.root.grad, Rt[i].grad), "b", err(l.bias.grad, bs[i].grad))
print("rel", err(dec.rel_emb.grad, rel.grad))
# the scorer alone on the SAME z (fp64 copy of the HIP z): is dz reproduced?
zz = z.detach().double().requires_grad_()
z16 = B.rf(zz)
s = torch.cat([(z16[e[0]] * rel.detach()[t] * z16[e[1]]).sum(1) for e, t in ((ei, et), (neg, nt))])
gt = torch.cat([torch.ones(ei.size(1), dtype=torch.float64, device=DEV), torch.zeros(neg.size(1), dtype=torch.float64, device=DEV)])
l2 = torch.nn.functional.binary_cross_entropy_with_logits(s, gt) + 1e-2 * zz.pow(2).mean()
l2.backward()
print("dz given the HIP z", err(z.grad, zz.grad), "scores", err(torch.cat([pp, npd]), s.detach()))
