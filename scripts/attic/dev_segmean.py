"""Dev harness for the R-GCN segment-mean gather (csrc/spmm.hip, tag rgcn_segment_mean): the bench graph's real
(relation, target) segments and uniform synthetic row lengths, bf16 table d = 256, HIP-event timings."""
import os
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import graph as G, ops, synthetic  # noqa: E402

ops.set_matmul_precision("bf16")
dev = "cuda"
N, D = 130_000, int(os.environ.get("SEGMEAN_D", "256"))
GCN = os.environ.get("SEGMEAN_GRAPH", "rgcn") == "gcn"     # the GCN propagate's by-target CSR instead of the segments


def t(fn, reps=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def reference(rowptr, col, w, x, n_rows):
    rows = torch.repeat_interleave(torch.arange(n_rows, device=dev), (rowptr[1:] - rowptr[:-1]).long())
    out = torch.zeros(n_rows, x.size(1), device=dev, dtype=torch.float64)
    out.index_add_(0, rows, x[col.long()].double() * w.double()[:, None])
    return out


def main():
    torch.manual_seed(0)
    x = torch.randn(N, D, device=dev).bfloat16()
    g = synthetic.primekg_like(N, 4_000_000, 30, seed=42)
    rg = G.RelGraph(g["edge_index"].to(dev), g["edge_type"].to(dev), N, g["num_relations"])
    if GCN:
        gr = G.Graph(g["edge_index"].to(dev), N)
        c = gr.by_src if os.environ.get("SEGMEAN_BWD") == "1" else gr.by_dst
        gw = gr.gcn_weights()

        class RG:
            seg_ptr, src_sorted, num_segments, seg_split = c.rowptr, c.col, N, c.split
            inv_cnt = gw.w_src if os.environ.get("SEGMEAN_BWD") == "1" else gw.w_dst
        rg = RG
    run = lambda: ops.spmm(rg.seg_ptr, rg.src_sorted, rg.inv_cnt, x, rg.num_segments, split=rg.seg_split, out_dtype=torch.bfloat16)  # noqa: E731
    out = run().float()
    ref = reference(rg.seg_ptr, rg.src_sorted, rg.inv_cnt, x, rg.num_segments)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    us = t(run)
    print("items:", rg.seg_split.totals.tolist(), "packed" if rg.seg_split.pack else "one row per item")
    sp = rg.seg_split
    keep = (sp.c.n_items_max, sp.c.n_long_max, sp.c.n_short_max)
    tot = sp.totals.tolist()
    sp.c.n_items_max, sp.c.n_long_max, sp.c.n_short_max = max(tot[0], 1), max(tot[2], 1), tot[3]
    print(f"real segments, exact grids: {t(run):.1f} us", flush=True)
    sp.c.n_items_max, sp.c.n_long_max, sp.c.n_short_max = keep
    by = rg.src_sorted.numel() * (D * 2 + 8) + rg.num_segments * (D * 2 + 4)
    print(f"real segments: S={rg.num_segments} {us:.1f} us  {by / us / 1e6:.2f} TB/s algorithmic  max rel err {err:.2e}", flush=True)
    cfgs = {}
    for chunk in (64, 128, 256):
        for pack in (True, False):
            cfgs[(chunk, pack)] = G.RowSplit(rg.seg_ptr, rg.num_segments, rg.src_sorted.numel(), chunk=chunk, pack=pack, col=rg.src_sorted)
    res = {k: [] for k in cfgs}
    for rnd in range(4):                      # round robin: clock / cache state drifts between measurements
        for k, sp2 in cfgs.items():
            res[k].append(t(lambda: ops.spmm(rg.seg_ptr, rg.src_sorted, rg.inv_cnt, x, rg.num_segments, split=sp2, out_dtype=torch.bfloat16), reps=20))
    for k, v in res.items():
        print(f"real segments, chunk={k[0]} pack={k[1]}: items={cfgs[k].totals.tolist()} " + " ".join(f"{u:.1f}" for u in v) + " us", flush=True)
    # ceiling of the REAL index stream: the same 4 M gathers in the same order, cut into uniform rows of 16 / 64 / 256
    # entries (no short rows, no long rows, no fix-up): what the memory system gives this access pattern
    for ln in (16, 64, 256):
        rows = rg.src_sorted.numel() // ln
        rowptr = (torch.arange(rows + 1, device=dev) * ln).to(torch.int32)
        for name, split in (("csr", None), ("items", G.RowSplit(rowptr, rows, rows * ln))):
            us = t(lambda: ops.spmm(rowptr, rg.src_sorted, rg.inv_cnt, x, rows, split=split, out_dtype=torch.bfloat16))
            by = rows * ln * (D * 2 + 8) + rows * (D * 2 + 4)
            print(f"real index stream, uniform rows of {ln} ({name}): {us:.1f} us  {by / us / 1e6:.2f} TB/s", flush=True)
    # the real segments by length class, each class alone (same sources, same order)
    lens = (rg.seg_ptr[1:] - rg.seg_ptr[:-1]).long()
    row_of_entry = torch.repeat_interleave(torch.arange(lens.numel(), device=dev), lens)
    for name, m in (("len<=8", lens <= 8), ("9..32", (lens > 8) & (lens <= 32)), ("33..256", (lens > 32) & (lens <= 256)),
                    ("len>256", lens > 256)):
        sel = m[row_of_entry]
        l2 = lens[m]
        rowptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(l2, 0)]).to(torch.int32)
        col, w = rg.src_sorted[sel].contiguous(), rg.inv_cnt[sel].contiguous()
        rows, nnz = int(l2.numel()), int(col.numel())
        for pack in (False, True):
            sp = G.RowSplit(rowptr, rows, nnz, pack=pack, col=col)
            us = t(lambda: ops.spmm(rowptr, col, w, x, rows, split=sp, out_dtype=torch.bfloat16))
            by = nnz * (D * 2 + 8) + rows * (D * 2 + 4)
            print(f"class {name}: rows={rows} entries={nnz} items={sp.totals.tolist()} pack={pack}: {us:.1f} us  {by / us / 1e6:.2f} TB/s", flush=True)
            tot = sp.totals.tolist()          # the same launch with grids cut to the exact counts
            sp.c.n_items_max, sp.c.n_long_max, sp.c.n_short_max = max(tot[0], 1), max(tot[2], 1), tot[3]
            us = t(lambda: ops.spmm(rowptr, col, w, x, rows, split=sp, out_dtype=torch.bfloat16))
            print(f"      exact grids: {us:.1f} us  {by / us / 1e6:.2f} TB/s", flush=True)
    if os.environ.get("SEGMEAN_CEILING") == "1":
        return
    # mixed row lengths (empties, short runs, medium, long) against the fp64 scatter-add, several widths / precisions
    gen = torch.Generator(device=dev).manual_seed(3)
    for d, dt in ((256, torch.bfloat16), (512, torch.bfloat16), (128, torch.bfloat16), (64, torch.bfloat16), (32, torch.bfloat16),
                  (256, torch.float32), (128, torch.float32), (1024, torch.bfloat16)):
        rows = 60_000
        lens = torch.randint(0, 6, (rows,), device=dev, generator=gen)
        lens[torch.rand(rows, device=dev, generator=gen) < 0.1] = 0
        big = torch.rand(rows, device=dev, generator=gen)
        lens[big < 0.05] = torch.randint(6, 40, (rows,), device=dev, generator=gen)[big < 0.05]
        lens[:3] = torch.tensor([700, 1, 300], device=dev)
        rowptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(lens, 0)]).to(torch.int32)
        nnz = int(rowptr[-1])
        col = torch.randint(0, N, (nnz,), device=dev, dtype=torch.int32)
        w = torch.rand(nnz, device=dev)
        xt = torch.randn(N, d, device=dev).to(dt)
        bias = torch.randn(d, device=dev)
        ref = reference(rowptr, col, w, xt.float(), rows)
        for pack in (False, True):
            sp = G.RowSplit(rowptr, rows, nnz, pack=pack, col=col)
            o = ops.spmm(rowptr, col, w, xt, rows, split=sp, out_dtype=torch.float32)
            err = float((o.double() - ref).abs().max() / ref.abs().max())
            o2 = ops.spmm(rowptr, col, w, xt, rows, split=sp, bias=bias, relu=True, out_dtype=torch.float32)
            err2 = float((o2.double() - torch.relu(ref + bias.double())).abs().max() / ref.abs().max())
            items = int(sp.totals[0])
            print(f"mixed d={d} {str(dt)[6:]} pack={pack}: items={items} err={err:.2e} err(bias,relu)={err2:.2e}", flush=True)
            assert err < 1e-5 and err2 < 1e-5
    if os.environ.get("SEGMEAN_ONLY_REAL") == "1":
        return
    for rows, ln in ((272_000, 1), (272_000, 2), (272_000, 4), (272_000, 8), (272_000, 16), (68_000, 16), (8192, 256), (16384, 256),
                     (2048, 2048)):
        rowptr = (torch.arange(rows + 1, device=dev) * ln).to(torch.int32)
        col = torch.randint(0, N, (rows * ln,), device=dev, dtype=torch.int32)
        w = torch.rand(rows * ln, device=dev)
        for name, split in (("csr", None), ("items", G.RowSplit(rowptr, rows, rows * ln))):
            us = t(lambda: ops.spmm(rowptr, col, w, x, rows, split=split, out_dtype=torch.bfloat16))
            by = rows * ln * (D * 2 + 8) + rows * (D * 2 + 4)
            print(f"rows={rows} len={ln} {name}: {us:.1f} us  {by / us / 1e6:.2f} TB/s  {us * 1e3 / rows:.2f} ns/row", flush=True)


main()
