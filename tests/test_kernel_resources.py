"""Compile-time properties the DMA-ring kernels depend on (no GPU needed: hipcc cross-compiles gfx950).

``gemm_xw8k_kernel`` (csrc/gemm_xw.hip) waits for its LDS-DMA ring with COUNTED ``s_waitcnt vmcnt(N)``: a register spill is
a scratch access on that same counter (and hipcc drains it with ``vmcnt(0)``), so a build whose K = 768 instantiation spills
-- 192 of its 256 registers hold weights -- is a wrong build, not a slow one."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "biomedkg_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_weight_slab_kernels_do_not_spill(tmp_path):
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
                          "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, "gemm_xw.hip"), "-o",
                          str(tmp_path / "x.o")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", out.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "gemm_xw8k_kernel" not in name:
            continue
        seen += 1
        spill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        sspill = int(re.search(r"SGPRs Spill: (\d+)", b).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vgprs = int(re.search(r" VGPRs: (\d+)", b).group(1))
        assert spill == 0 and sspill == 0 and scratch == 0 and vgprs <= 256, (name, vgprs, spill, sspill, scratch)
    assert seen == 6, seen          # {bf16, fp32} x {K = 512, 768} plain + the two ReDAF instantiations


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_gated_product_and_one_pass_walk_do_not_spill(tmp_path):
    """``gemm_xw8_kernel<bf16, GATED>`` issues its gate loads by hand and counts them into the ring's ``vmcnt`` ladder -- a
    spill would be one more memory operation on that counter; ``kge_onepass_kernel`` at the bench width (bf16, 16 lanes per
    row) keeps four 16-register vectors and two batches of rows in flight: it must fit three waves per SIMD without scratch."""
    # (round 6) ``kge_onepass_mx_kernel``: 256 registers and 74 KB of LDS are what puts two of its workgroups on a CU -- two
    # waves per SIMD hide each other's LDS and transcendental latencies; a spill would also sit on the row ring's vmcnt count
    for src, want in (("gemm_xw.hip", ("gemm_xw8_kernelIDF16bLb1E",)), ("kge_bwd_nodes.hip", ("kge_onepass_kernelIDF16bLi16ELi2E",)),
                      ("kge_onepass_mx.hip", ("kge_onepass_mx_kernelILb0E", "kge_onepass_mx_kernelILb1E"))):
        out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
                              "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", str(tmp_path / "x.o")],
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        blocks = re.split(r"remark: Function Name: ", out.stderr)[1:]
        for key in want:
            hit = [b for b in blocks if key in b.split()[0]]
            assert len(hit) == 1, (key, len(hit))
            b = hit[0]
            spill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
            scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
            occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
            assert spill == 0 and scratch == 0 and occ >= 2, (key, spill, scratch, occ)
