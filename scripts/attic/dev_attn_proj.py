import sys, time, torch
sys.path.insert(0, ".")
from biomedkg_amd import ops, synthetic
from biomedkg_amd.utils.fusion import AttentionFusion
from oracle import ref_modules as O
DEV = "cuda"
for d, n, big_bias in ((768, 6000, True), (768, 6000, False), (256, 6000, True)):
    x = synthetic.lm_features(n, 2, d, seed=3)
    torch.manual_seed(2)
    f = AttentionFusion(d).to(DEV).eval()
    if big_bias:
        with torch.no_grad():
            for p in (f.q_proj.bias, f.k_proj.bias, f.v_proj.bias):
                p.uniform_(-0.3, 0.3)
    o = O.AttentionFusion(d).double()
    o.load_state_dict({k: v.double().cpu() for k, v in f.state_dict().items()})
    with torch.no_grad():
        want = o(x.double())
        want32 = O.AttentionFusion(d)
        want32.load_state_dict({k: v.cpu() for k, v in f.state_dict().items()})
        w32 = want32(x).double()
    xd = x.to(DEV)
    print("precision", ops.get_matmul_precision())
    exact = f(xd).double().cpu()
    rel = lambda a: float((a - want).norm() / want.norm())
    print(d, big_bias, "oracle fp32 vs fp64", rel(w32), "hip fp32", rel(exact))
    ops.set_matmul_precision("bf16")
    got = f(xd.clone()).double().cpu()
    ops.PROJECTED_ATTENTION = False
    lit = f(xd.clone()).double().cpu()
    ops.PROJECTED_ATTENTION = True
    ops.set_matmul_precision("fp32")
    print("   bf16 projected", rel(got), "bf16 literal", rel(lit))
# timing at full size
n = 130_000
x = synthetic.lm_features(n, 2, 768, seed=3).to(DEV)
f = AttentionFusion(768).to(DEV).eval()
ops.set_matmul_precision("bf16")
for proj in (True, False):
    ops.PROJECTED_ATTENTION = proj
    for _ in range(2):
        x.add_(0.0); f(x)
    ts = []
    for _ in range(5):
        x.add_(0.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f(x)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("projected" if proj else "literal", "first-call ms", sorted(ts))
