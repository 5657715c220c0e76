"""Dev: the two-view GCN propagate (d = 512 bf16) as ONE launch over 1 KB rows against TWO launches over its 256-column
halves (512-byte row pieces of the same table, ldx = ldo = 512)."""
import sys, torch
sys.path.insert(0, ".")
from biomedkg_amd import ops, synthetic, graph as G, _lib as L
ops.set_matmul_precision("bf16")
dev = "cuda"
g = synthetic.primekg_like(130_000, 4_000_000, 30, seed=42)
ei = g["edge_index"].to(dev)
gr = G.Graph(ei, 130_000)
gw = gr.gcn_weights()
c = gr.by_dst
n = 130_000
x = torch.randn(n, 512, device=dev).bfloat16()
out = torch.empty(n, 512, dtype=torch.bfloat16, device=dev)
for _ in range(4):
    c.split.use()


def call(xv, ov, d, halves=1):
    sp = c.split
    if d // halves // 8 >= 64 and sp.exact and sp.pack:
        sp = sp.plain()
    partials = sp.partials(d, dev)
    L.call("bkg_spmm_csr_f32", L.ptr(c.rowptr), L.ptr(c.col), L.ptr(gw.w_dst), L.ptr(xv), 1, x.stride(0), L.ptr(ov), 1, out.stride(0),
           n, d, L.ptr(gw.self_w), None, 0, 0, None, 0, None, 0.0, 0, 0, L.sref(sp.c), L.ptr(partials), halves, L.stream())


def t(fn, reps=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


one = t(lambda: call(x, out, 512))
ref = out.clone()
def halves():
    call(x[:, :256], out[:, :256], 256)
    call(x[:, 256:], out[:, 256:], 256)
out.zero_()
two = t(halves)
two_out = out.clone()
out.zero_()
xcd = t(lambda: call(x, out, 512, 2))
print("one launch", round(one, 1), "us; two launches over the halves", round(two, 1), "us; one launch, halves by XCD", round(xcd, 1),
      "us; halves equal", bool(torch.equal(two_out, out)), "max diff vs whole rows",
      float((ref.float() - out.float()).abs().max()), float(ref.float().abs().max()))

xcd4 = t(lambda: call(x, out, 512, 4))
print("one launch, quarters by XCD pair", round(xcd4, 1), "us; max diff", float((ref.float() - out.float()).abs().max()))
x2 = x[:, :256].contiguous()
out2 = torch.empty(n, 256, dtype=torch.bfloat16, device=dev)
def call2(halves):
    sp = c.split
    partials = sp.partials(256, dev)
    L.call("bkg_spmm_csr_f32", L.ptr(c.rowptr), L.ptr(c.col), L.ptr(gw.w_dst), L.ptr(x2), 1, x2.stride(0), L.ptr(out2), 1, out2.stride(0),
           n, 256, L.ptr(gw.self_w), None, 0, 0, None, 0, None, 0.0, 0, 0, L.sref(sp.c), L.ptr(partials), halves, L.stream())
a1 = t(lambda: call2(1)); r2 = out2.clone()
a2 = t(lambda: call2(2))
print("d = 256: one launch", round(a1, 1), "us; halves by XCD", round(a2, 1), "us; max diff", float((r2.float() - out2.float()).abs().max()))

xf = x.float()
outf = torch.empty(n, 512, device=dev)
def callf(halves):
    sp = c.split
    if 512 // halves // 4 >= 64 and sp.exact and sp.pack:
        sp = sp.plain()
    partials = sp.partials(512, dev)
    L.call("bkg_spmm_csr_f32", L.ptr(c.rowptr), L.ptr(c.col), L.ptr(gw.w_dst), L.ptr(xf), 0, xf.stride(0), L.ptr(outf), 0, outf.stride(0),
           n, 512, L.ptr(gw.self_w), None, 0, 0, None, 0, None, 0.0, 0, 0, L.sref(sp.c), L.ptr(partials), halves, L.stream())
f1 = t(lambda: callf(1)); rf = outf.clone()
f2 = t(lambda: callf(2)); d2 = float((rf - outf).abs().max())
f4 = t(lambda: callf(4)); d4 = float((rf - outf).abs().max())
print("fp32 d = 512: one launch", round(f1, 1), "us; halves", round(f2, 1), "us; quarters", round(f4, 1), "us; max diff", d2, d4)
