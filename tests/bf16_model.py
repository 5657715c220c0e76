"""fp64 restatement of the bf16 training path WITH its rounding points (test infrastructure; evaluated on the GPU in
float64 by tests/test_hip_configs.py at BASELINE configs[2]'s full size).

bf16 mode ("bf16 storage / fp32 accumulate") stores as bf16 exactly the tensors that are only ever an
MFMA operand, a gather table or a ReLU gate: the input features, every weight image, the products ``x W^T`` (gather
tables), the R-GCN segment means and per-relation products, the hidden activations between conv layers -- and, in the
backward, the gated output gradient of a layer, the table gradients ``A^T dOut`` / ``dH``, and the gradients of the
hidden activations.  Everything else (accumulations, the encoder output, scorer, losses, weight gradients) is fp32.

This module restates the two timed steps (GGD over the GCN encoder with both views side by side; R-GCN + DistMult with
the fused BCE + L2 loss) in float64 torch autograd and rounds to bf16 at THOSE points -- ``rf`` rounds a value
(straight-through gradient), ``rb`` rounds the gradient that flows back through it -- so that the difference to the HIP
path is accumulation order / fp32-vs-fp64 accumulation only, not bf16 rounding.  The arithmetic between the rounding
points follows the reference (biomedkg/model/encoder.py:50-59,153-162, gcl.py:73-93, kge_module.py:109-129) and PyG's
GCNConv / RGCNConv semantics."""
import torch


def bf(x):
    return x.float().bfloat16().double()


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return bf(g)


rf, rb = _RoundFwd.apply, _RoundBwd.apply


class _Aggregate(torch.autograd.Function):
    """``out[dst_e] += w_e * table[src_e]`` (+ ``self_w[i] * table[i]``) in float64, edges in chunks; the backward is the
    transposed sum.  ``n_out`` rows out, any number of table rows."""

    CHUNK = 1 << 19

    @staticmethod
    def forward(ctx, table, src, dst, w, self_w, n_out):
        out = torch.zeros(n_out, table.size(1), dtype=torch.float64, device=table.device)
        for a in range(0, src.numel(), _Aggregate.CHUNK):
            s = slice(a, a + _Aggregate.CHUNK)
            out.index_add_(0, dst[s], table[src[s]] * w[s, None])
        if self_w is not None:
            out += table[:n_out] * self_w[:, None]
        ctx.save_for_backward(src, dst, w, self_w)
        ctx.n_table = table.size(0)
        return out

    @staticmethod
    def backward(ctx, g):
        src, dst, w, self_w = ctx.saved_tensors
        d = torch.zeros(ctx.n_table, g.size(1), dtype=torch.float64, device=g.device)
        for a in range(0, src.numel(), _Aggregate.CHUNK):
            s = slice(a, a + _Aggregate.CHUNK)
            d.index_add_(0, src[s], g[dst[s]] * w[s, None])
        if self_w is not None:
            d[: g.size(0)] += g * self_w[:, None]
        return d, None, None, None, None, None


def aggregate(table, src, dst, w, self_w, n_out):
    return _Aggregate.apply(table, src, dst, w, self_w, n_out)


def gcn_norm(edge_index, n):
    """PyG gcn_norm with add_remaining_self_loops, float64: (src, dst, w) of the non-loop edges and the
    unit self loop's weight per node."""
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    src, dst = src[keep], dst[keep]
    deg = torch.bincount(dst, minlength=n).double() + 1.0
    dis = deg.pow(-0.5)
    return src, dst, dis[src] * dis[dst], dis * dis


def leaf(p):
    return p.detach().double().requires_grad_()


def ggd_step(x, edge_index, perm, conv_w, conv_b, head_w, head_b, fold=True):
    """GGD's loss (reference gcl.py:73-93 without the augmentation branch, gcl_module.py:229-234) through the side-by-side
    bf16 path of model/gcl.py ``GGD._forward_views``.  ``conv_w[i] [out, in]`` / ``conv_b[i]``: float64 leaves.
    ``fold`` (the product's default since round 5): the last conv layer folded into the linear head -- the hidden
    activations H (bf16, their gradient a bf16 table) meet ``w' = W_L^T (W_m^T 1)`` in a dot product per row and ONE column
    per view goes through the aggregation, in fp32: no bf16 rounding of that layer's product, output or gradient tables."""
    n = x.size(0)
    src, dst, w, self_w = gcn_norm(edge_index, n)
    x16 = bf(x)                                                     # cached bf16 copy of the constant input features
    xw = rb(rf(x16 @ rf(conv_w[0]).t()))                            # shared first-layer product, a bf16 table
    table = rb(torch.cat([xw, xw[perm]], dim=1))                    # node i: [pos_i | neg_i]; its gradient is a bf16 table
    last = len(conv_w) - 1
    h = None
    for i in range(len(conv_w)):
        if fold and i == last and last >= 1:
            break
        if i > 0:
            d_in, d_out = conv_w[i].size(1), conv_w[i].size(0)
            table = rb(rf(h.view(2 * n, d_in) @ rf(conv_w[i]).t())).view(n, 2 * d_out)
        pre = rb(aggregate(table, src, dst, w, self_w, n) + conv_b[i].repeat(2))   # gated dOut is cast to bf16 in the backward
        h = rb(rf(torch.relu(pre) if i != last else pre))           # hidden activations and the head's operand: bf16
    if fold and last >= 1:
        v = head_w.sum(0)
        w_fold = conv_w[last].t() @ v
        u = (h.view(2 * n, conv_w[last].size(1)) @ w_fold).view(n, 2)
        s = aggregate(u, src, dst, w, self_w, n) + (conv_b[last] @ v + head_b.sum())
    else:
        hh = h.view(2 * n, h.size(1) // 2)
        s = (hh @ head_w.sum(0) + head_b.sum()).view(n, 2)          # (W h + b).sum(1) = h . W.sum(0) + b.sum()
    pred = torch.cat([s[:, 0], s[:, 1]])
    gt = torch.cat([torch.ones(n, dtype=torch.float64, device=x.device), torch.zeros(n, dtype=torch.float64, device=x.device)])
    return torch.nn.functional.binary_cross_entropy_with_logits(pred, gt)


GATE_MISMATCHES = []     # per gated layer: (elements gated differently by `gate` and by this model's own sign, elements)


def rgcn_layer(x, src, dst, seg_of_edge, seg_dst, seg_rel, inv_cnt, weight, root, bias, n, relu, hidden, gate=None):
    """One RGCNConv (PyG aggr="mean", root_weight, bias) + the encoder's ReLU, rounding where ops._RgcnConv
    does.  ``x``: the layer input as the kernels see it (bf16 values).  ``gate`` (0 / 1, optional): the ReLU decisions of
    the run this is compared with.  An element whose pre-activation lies within rounding noise of zero is gated one way
    by one side and the other way by the other -- both are right -- and ONE such element per 10^5 moves a gradient that is
    a sum of cancelling terms by 1e-3; with the decisions taken from the other run what is compared is the arithmetic.
    How many decisions differ from this model's own is recorded in ``GATE_MISMATCHES`` for the caller to bound."""
    n_seg = seg_dst.numel()
    ones = torch.ones(n_seg, dtype=torch.float64, device=x.device)
    # segment means, a bf16 table whose gradient dH is one too; the gradient of the TABLE part of x is not rounded by
    # itself (the source-side reduce adds the root part in its epilogue and rounds the sum: rb(rf(out)) of the layer below)
    h = rb(rf(aggregate(x, src, seg_of_edge, inv_cnt, None, n_seg)))
    w16 = rf(weight)
    counts = torch.bincount(seg_rel, minlength=weight.size(0)).tolist()             # segments are sorted by (relation, target)
    parts, a = [], 0
    for r, c in enumerate(counts):                                                  # per-relation products
        if c:
            parts.append(h[a:a + c] @ w16[r])
        a += c
    y = rf(torch.cat(parts))                                                        # read only as a gather table: bf16
    xr = rb(x) @ rf(root)                                                           # root term; its input gradient is a bf16 table
    if hidden:
        xr = rf(xr)                                                                 # a bf16 addend of the epilogue
    iota = torch.arange(n_seg, device=x.device)
    pre = rb(aggregate(y, iota, seg_dst, ones, None, n) + xr + bias)                # the (gated) dOut is cast to bf16
    if relu and gate is not None:
        GATE_MISMATCHES.append((int(((pre.detach() > 0) != (gate > 0)).sum()), pre.numel()))
        out = pre * gate
    else:
        out = torch.relu(pre) if relu else pre
    return rb(rf(out)) if hidden else out


def kge_step(x, edge_index, edge_type, neg_edge_index, neg_edge_type, weights, roots, biases, rel_emb, num_relations,
             gates=None):
    """``KGEModule.training_step`` (reference kge_module.py:109-129): R-GCN stack, DistMult scores of the positives and the
    given negatives, BCE with logits + 1e-2 (mean z^2 + mean rel_emb^2), through bf16 mode's rounding points."""
    n = x.size(0)
    src, dst = edge_index[0], edge_index[1]
    key = edge_type * n + dst
    uniq, seg_of_edge, cnt = torch.unique(key, return_inverse=True, return_counts=True)
    seg_dst, seg_rel = uniq % n, uniq // n
    inv_cnt = 1.0 / cnt[seg_of_edge].double()
    h = bf(x)
    last = len(weights) - 1
    for i in range(len(weights)):
        h = rgcn_layer(h, src, dst, seg_of_edge, seg_dst, seg_rel, inv_cnt, weights[i], roots[i], biases[i], n,
                       relu=(i != last), hidden=(i != last), gate=None if (gates is None or i == last) else gates[i])
    z = h
    z16 = rf(z)                                                     # the scorer gathers node rows from a bf16 copy of z

    def chunk_scores(zz, rel, h, t, r):
        return (zz[h] * rel[r] * zz[t]).sum(1)

    def scores(ei, et):
        from torch.utils.checkpoint import checkpoint
        out = []
        for a in range(0, ei.size(1), 1 << 19):
            s = slice(a, a + (1 << 19))
            out.append(checkpoint(chunk_scores, z16, rel_emb, ei[0, s], ei[1, s], et[s], use_reentrant=False))
        return torch.cat(out)
    pred = torch.cat([scores(edge_index, edge_type), scores(neg_edge_index, neg_edge_type)])
    gt = torch.cat([torch.ones(edge_index.size(1), dtype=torch.float64, device=x.device),
                    torch.zeros(neg_edge_index.size(1), dtype=torch.float64, device=x.device)])
    ce = torch.nn.functional.binary_cross_entropy_with_logits(pred, gt)
    return ce + 1e-2 * (z.pow(2).mean() + rel_emb.pow(2).mean()), z
