"""Per-phase wall times of the bench step at a given scale (diagnostic; GPU only)."""
import sys, time, torch
sys.path.insert(0, ".")
from biomedkg_amd import ops, synthetic, graph as G, sampling
from biomedkg_amd.kge_module import KGEModule
from biomedkg_amd.gcl_module import GGDModule

n, e = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda"
def T(msg, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{msg:40s} {1e3*(time.perf_counter()-t):10.2f} ms", flush=True); return r
g = T("gen", lambda: synthetic.primekg_like(n, e, 30, seed=42))
ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
gr = T("Graph build", lambda: G.Graph(ei, n))
T("gcn_weights", lambda: gr.gcn_weights())
rg = T("RelGraph build", lambda: G.RelGraph(ei, et, n, 30))
print("segments", rg.num_segments)
x = torch.randn(n, 256, device=dev)
for rep in range(2):
    h = T("spmm seg mean", lambda: ops.spmm(rg.seg_ptr, rg.src_sorted, rg.inv_cnt, x, rg.num_segments))
    w = torch.randn(30, 256, 256, device=dev) * 0.05
    y = T("grouped rows gemm", lambda: ops.gemm_grouped_rows(h, w, rg.rel_ptr))
    T("grouped rows gemm^T", lambda: ops.gemm_grouped_rows(y, w, rg.rel_ptr, trans_b=True))
    T("grouped k gemm", lambda: ops.gemm_grouped_k(h, y, rg.rel_ptr, 30))
    T("spmm dst reduce", lambda: ops.spmm(rg.dst_csr.rowptr, rg.dst_csr.col, None, y, n))
    T("spmm src (bwd)", lambda: ops.spmm(rg.src_csr.rowptr, rg.src_csr.col, rg.src_w, h, n))
    T("gemm root", lambda: ops.gemm(x, w[0]))
    T("gemm dW", lambda: ops.gemm(x, x, trans_a=True))
    neg = T("neg sampling", lambda: sampling.negative_sampling(ei, num_neg_samples=e))
    rel = torch.randn(30, 256, device=dev)
    s = T("score fwd", lambda: ops.kge_score(x, rel, ei, et, 0))
    inc = T("incidence", lambda: ops.Incidence(neg, n))
    xr, rr = x.clone().requires_grad_(True), rel.clone().requires_grad_(True)
    s = ops.kge_score(xr, rr, neg, et, 0)
    T("score bwd (neg, incl incidence)", lambda: s.backward(torch.ones_like(s)))
    c = gr.by_dst; gw = gr.gcn_weights()
    T("gcn spmm", lambda: ops.spmm(c.rowptr, c.col, gw.w_dst, x, n, self_w=gw.self_w))
print("phases ok")
class B: pass
bk = B(); bk.x, bk.edge_index, bk.edge_type = x, ei, et
kge = KGEModule("rgcn", "dismult", 256, 256, 256, 2, 30, 2, "cosine", 1e-3, 0.2, "none", 1, "random").to(dev).train()
opt = torch.optim.Adam(kge.model.parameters(), lr=1e-3)
for i in range(3):
    def st():
        opt.zero_grad(); l = kge.training_step(bk); l.backward(); torch.nn.utils.clip_grad_norm_(kge.model.parameters(), 1.0); opt.step(); return l
    l = T("kge step", st)
print("loss", l.item())
bg = B(); bg.x, bg.edge_index = torch.randn(n, 768, device=dev), ei
gcl = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(dev).train()
og = torch.optim.Adam(gcl.model.parameters(), lr=1e-3)
for i in range(3):
    def st():
        og.zero_grad(); l = gcl.training_step(bg); l.backward(); torch.nn.utils.clip_grad_norm_(gcl.model.parameters(), 1.0); og.step(); return l
    l = T("gcl step", st)
print("loss", l.item())
