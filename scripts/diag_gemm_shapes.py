#!/usr/bin/env python3
"""Which products does one bench step launch, with what operand types?  Wraps the library call of every GEMM entry point
for one GGD step and one KGE step (bf16 mode) and prints shape / dtype per call.  python scripts/diag_gemm_shapes.py"""
import sys
import torch

sys.path.insert(0, ".")
import bench                                   # noqa: E402
from biomedkg_amd import _lib as L, ops        # noqa: E402

ops.set_matmul_precision("bf16")
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras"]
args = bench.parse()
dev = torch.device("cuda:0")
gstep, kstep = bench.build_replica_steps(args, 0, 1, dev)[:2]
for _ in range(2):
    gstep(); kstep()
torch.cuda.synchronize()
real = L.call
names = ("bkg_gemm_xw", "bkg_gemm_tn_bf16", "bkg_gemm_grouped_k_bf16", "bkg_gemm_f32", "bkg_gemm_grouped_rows_f32",
         "bkg_gemm_grouped_k_f32")


def spy(name, *a):
    if name in names:
        if name == "bkg_gemm_xw":
            print(f"{name}: M={a[0]} N={a[1]} K={a[2]} A={'bf16' if a[5] else 'f32'} rows={'y' if a[7] else 'n'} "
                  f"groups={a[12]} C={'bf16' if a[15] else 'f32'} bias={'y' if a[17] else 'n'} act={a[18]} gate={'y' if a[19] else 'n'}")
        elif name == "bkg_gemm_tn_bf16":
            print(f"{name}: M={a[0]} N={a[1]} K={a[2]} A={'bf16' if a[4] else 'f32'} B={'bf16' if a[7] else 'f32'}")
        elif name == "bkg_gemm_grouped_k_bf16":
            print(f"{name}: rows={a[0]} M={a[1]} N={a[2]} A={'bf16' if a[4] else 'f32'} B={'bf16' if a[7] else 'f32'} b_rows={'y' if a[9] else 'n'}")
        else:
            print(f"{name}: {[x for x in a[:6] if isinstance(x, (int, float))]}")
    return real(name, *a)


L.call = spy
ops.L.call = spy
print("---- GGD step"); gstep(); torch.cuda.synchronize()
print("---- KGE step"); kstep(); torch.cuda.synchronize()
