"""A torch (CPU) arithmetic backend for biomedkg_amd.partition -- TEST INFRASTRUCTURE ONLY: it lets the gloo tests
exercise the partition / all-gather / reduce-scatter logic without a GPU.  The product always passes HipBackend."""
import torch
import torch.nn.functional as F


class TorchCSR:
    split = None

    def __init__(self, key, other, n_rows):
        key, other = key.long(), other.long()
        self.perm = torch.argsort(key, stable=True)
        self.col = other[self.perm]
        self.rowptr = torch.searchsorted(key[self.perm].contiguous(), torch.arange(n_rows + 1))
        self.n_rows = n_rows

    def rows(self):
        return torch.repeat_interleave(torch.arange(self.n_rows), self.rowptr[1:] - self.rowptr[:-1])


def _spmm(csr, w, table, n_out):
    vals = table[csr.col]
    if w is not None:
        vals = vals * w[:, None]
    return torch.zeros(n_out, table.size(1), dtype=table.dtype).index_add_(0, csr.rows(), vals)


class _SpmmFn(torch.autograd.Function):
    """Mirrors biomedkg_amd.ops._SpmmFn: the backward really goes through the TRANSPOSED CSR."""

    @staticmethod
    def forward(ctx, table, addend, bias, fwd, bwd, w_fwd, w_bwd, n_out, relu, keep, scale):
        out = _spmm(fwd, w_fwd, table, n_out)
        if addend is not None:
            out = out + addend
        if bias is not None:
            out = out + bias
        if relu:
            out = out.relu()
        if keep is not None:
            out = out * keep * scale
        ctx.bwd, ctx.w_bwd, ctx.n_table, ctx.scale, ctx.gated = bwd, w_bwd, table.size(0), scale, relu or keep is not None
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        if ctx.gated:
            dout = torch.where(out > 0, dout * ctx.scale, torch.zeros(()))
        need = ctx.needs_input_grad
        return (_spmm(ctx.bwd, ctx.w_bwd, dout, ctx.n_table) if need[0] else None, dout if need[1] else None,
                dout.sum(0) if need[2] else None, None, None, None, None, None, None, None, None)


class TorchBackend:
    csr = staticmethod(lambda key, other, n_rows: TorchCSR(key, other, n_rows))
    permute = staticmethod(lambda csr, values: values[csr.perm])
    linear = staticmethod(lambda x, weight, bias=None: F.linear(x, weight, bias))
    matmul = staticmethod(lambda x, w: x @ w)

    @staticmethod
    def spmm_fn(table, fwd, bwd, w_fwd, w_bwd, n_out, addend=None, bias=None, relu=False, drop=None):
        keep = drop.keep if drop is not None else None
        scale = 1.0 / (1.0 - drop.p) if drop is not None else 1.0
        return _SpmmFn.apply(table, addend, bias, fwd, bwd, w_fwd, w_bwd, n_out, relu, keep, scale)

    @staticmethod
    def grouped_linear(h, weight, group_ptr):
        ptr = group_ptr.tolist()
        return torch.cat([h[a:b] @ weight[g] for g, (a, b) in enumerate(zip(ptr[:-1], ptr[1:]))], dim=0)
