"""A torch (CPU) arithmetic backend for biomedkg_amd.partition -- TEST INFRASTRUCTURE ONLY: it lets the gloo tests
exercise the partition / halo-exchange logic without a GPU.  The product always passes HipBackend."""
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


class TorchBackend:
    fused = False            # the composable form of biomedkg_amd/partition.py (the product's HipBackend runs the fused layers)
    csr = staticmethod(lambda key, other, n_rows: TorchCSR(key, other, n_rows))
    rel_graph = staticmethod(lambda *a, **k: None)

    @staticmethod
    def gcn_norm(edge_index, num_nodes, keep):
        """PyG gcn_norm in edge order: (w_edge [E], self_w [N]); existing self loops get weight 0."""
        src, dst = edge_index[0], edge_index[1]
        alive = src != dst
        if keep is not None:
            alive = alive & keep.to(torch.bool)
        deg = torch.bincount(dst[alive], minlength=num_nodes).to(torch.float32) + 1.0
        dis = deg.pow(-0.5)
        return dis[src] * dis[dst] * alive.to(torch.float32), dis * dis

    permute = staticmethod(lambda csr, values: values[csr.perm])
    linear = staticmethod(lambda x, weight, bias=None: F.linear(x, weight, bias))
    matmul = staticmethod(lambda x, w: x @ w)
    matmul_tn = staticmethod(lambda a, b: a.t() @ b)
    matmul_nt = staticmethod(lambda a, b: a @ b.t())

    @staticmethod
    def spmm_epilogue(csr, w, table, n_out, addend, bias, relu, drop):
        out = _spmm(csr, w, table, n_out)
        if addend is not None:
            out = out + addend
        if bias is not None:
            out = out + bias
        if relu:
            out = out.relu()
        if drop is not None:
            out = out * drop.keep / (1.0 - drop.p)
        return out

    @staticmethod
    def gate(dout, out, relu, drop):
        scale = 1.0 / (1.0 - drop.p) if drop is not None else 1.0
        if relu or drop is not None:
            return torch.where(out > 0, dout * scale, torch.zeros(()))
        return dout

    @staticmethod
    def grouped_linear(h, weight, group_ptr):
        ptr = group_ptr.tolist()
        return torch.cat([h[a:b] @ weight[g] for g, (a, b) in enumerate(zip(ptr[:-1], ptr[1:]))], dim=0)

    @staticmethod
    def grouped_outer(h, dy, group_ptr, num_groups):
        ptr = group_ptr.tolist()
        return torch.stack([h[a:b].t() @ dy[a:b] for a, b in zip(ptr[:-1], ptr[1:])])

    @staticmethod
    def gcn_conv(x_local, weight, bias, pg, gw, relu=False, drop=None):
        from biomedkg_amd.partition import gcn_conv_partitioned
        return gcn_conv_partitioned(x_local, weight, bias, pg, TorchBackend, relu, drop, gw)

    @staticmethod
    def table_conv(table, bias, pg, gw, relu=False, drop=None):
        from biomedkg_amd.partition import _TableConv
        return _TableConv.apply(table, bias, pg.comm, TorchBackend, pg.by_dst, pg.by_src, gw.w_dst, gw.w_src, relu, drop)


class TorchNtXentKernels:
    """Dense torch restatement of the four NT-Xent kernels ``partition.ntxent_partitioned`` calls (test infrastructure)."""

    def normalize(self, h1, h2):
        z = torch.cat([h1, h2])
        norm = z.norm(dim=1).clamp_min(1e-12)
        zn = z / norm[:, None]
        n = h1.size(0)
        return zn, norm, (zn[:n] * zn[n:]).sum(1)

    def row_sums(self, z_all, anchors, tau):
        a = anchors.long()
        e = torch.exp(z_all[a] @ z_all.t() / tau)
        e[torch.arange(a.numel()), a] = 0.0
        return e.sum(1)

    def row_grads(self, z_all, anchors, den_all, norm, tau, scale, n_half):
        a = anchors.long()
        m = z_all.size(0)
        e = torch.exp(z_all[a] @ z_all.t() / tau)
        e[torch.arange(a.numel()), a] = 0.0
        w = e * (1.0 / den_all[a][:, None] + 1.0 / den_all[None, :])
        dzn = (w @ z_all - 2.0 * z_all[(a + m // 2) % m]) * scale
        zi = z_all[a]
        return (dzn - zi * (zi * dzn).sum(1, keepdim=True)) / norm[:, None]
