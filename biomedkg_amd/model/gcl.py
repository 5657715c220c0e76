"""GCL wrappers behind the reference's class names (``biomedkg/model/gcl.py``)."""
from __future__ import annotations

import math

import torch

from .. import ops
from .. import rand as _rand


class DGI(torch.nn.Module):
    """reference gcl.py:8-27."""

    def __init__(self, encoder, hidden_dim):
        super().__init__()
        self.encoder = encoder
        self.project = torch.nn.Linear(hidden_dim, hidden_dim)
        bound = 1.0 / math.sqrt(hidden_dim)  # PyG inits.uniform(size, weight)
        with torch.no_grad():
            self.project.weight.uniform_(-bound, bound)

    @staticmethod
    def corruption(x, edge_index):
        return x[_rand.current().perm(x.size(0), x.device)], edge_index

    @staticmethod
    def summary(z):
        return torch.sigmoid(z.mean(dim=0, keepdim=True))

    def forward(self, x, edge_index, summary=None):
        """``summary``: the readout in place of ``self.summary`` (a padded mini-batch averages its real rows only,
        biomedkg_amd/sampled_gcl.py)."""
        summary = summary or self.summary
        if ops.features_bf16_ok(x) and x.is_cuda:
            x = ops.cached_bf16(x)
        r = _rand.current()
        enc = self.encoder
        if (getattr(r, "device_draws", False) and x.is_cuda and x.dim() == 2 and not x.requires_grad
                and hasattr(enc, "forward_views") and x.size(1) % 256 == 0 and enc.views_ok(x.size(0))):
            # both passes side by side, the first layer's product shared (see GGD._views_ok)
            xw = ops.xw_table(x, enc.graph_layers[0].lin.weight)
            h2 = enc.forward_views(ops.pair_rows(xw, r.perm(x.size(0), x.device)), edge_index, 2)
            d = h2.size(1) // 2
            z, zn = h2[:, :d], h2[:, d:]
            return z, ops.linear(summary(z), self.project.weight, self.project.bias), zn
        z = self.encoder(x, edge_index)
        g = ops.linear(summary(z), self.project.weight, self.project.bias)
        zn = self.encoder(*self.corruption(x, edge_index))
        return z, g, zn


class GRACE(torch.nn.Module):
    """reference gcl.py:31-51.  ``forward`` keeps the reference's three encoder passes (the first is
    unused by the loss); ``views`` computes only the two the loss consumes."""

    def __init__(self, encoder, hidden_dim, proj_dim):
        super().__init__()
        self.encoder = encoder
        self.fc1 = torch.nn.Linear(hidden_dim, proj_dim)
        self.fc2 = torch.nn.Linear(proj_dim, hidden_dim)

    def _augment(self, x, edge_index):
        r = _rand.current()
        x1, x2 = r.masked(x, 0.4), r.masked(x, 0.4)
        e1 = r.edge_keep(edge_index.size(1), 0.4, x.device)
        e2 = r.edge_keep(edge_index.size(1), 0.4, x.device)
        return (x1, e1), (x2, e2)

    def forward(self, x, edge_index):
        (x1, e1), (x2, e2) = self._augment(x, edge_index)
        z = self.encoder(x, edge_index)
        return z, self.encoder(x1, edge_index, edge_keep=e1), self.encoder(x2, edge_index, edge_keep=e2)

    def views(self, x, edge_index):
        (x1, e1), (x2, e2) = self._augment(x, edge_index)
        if self.encoder.training and self.encoder.drop_out:
            _rand.current().skip(len(self.encoder.graph_layers) - 1)  # the unused pass's dropout draws
        return self.encoder(x1, edge_index, edge_keep=e1), self.encoder(x2, edge_index, edge_keep=e2)

    def project(self, z):
        return ops.linear(ops.linear(z, self.fc1.weight, self.fc1.bias, act=ops.ACT_ELU), self.fc2.weight, self.fc2.bias)


class GGD(torch.nn.Module):
    """reference gcl.py:54-93 (https://arxiv.org/abs/2206.01535)."""

    def __init__(self, encoder, hidden_dim, n_proj, aug_p):
        super().__init__()
        self.encoder = encoder
        self.p = aug_p
        self.mlp = torch.nn.ModuleList(torch.nn.Linear(hidden_dim, hidden_dim) for _ in range(n_proj))

    @staticmethod
    def corruption(x, edge_index):
        return x[_rand.current().perm(x.size(0), x.device)], edge_index

    def forward(self, x, edge_index):
        r = _rand.current()
        edge_keep = None
        if r.coin() < self.p:
            x = r.masked(x, 0.4)
            edge_keep = r.edge_keep(edge_index.size(1), 0.4, x.device)
        elif ops.features_bf16_ok(x) and x.is_cuda:
            x = ops.cached_bf16(x)      # bf16 mode: the corruption below then permutes a bf16 table
        if self._views_ok(x, r):
            return self._forward_views(x, edge_index, edge_keep, r)
        pos_h = self.encoder(x, edge_index, edge_keep=edge_keep)
        # the reference corrupts over the (possibly edge-dropped) graph of the positive pass
        neg_h = self.encoder(self.corruption(x, edge_index)[0], edge_index, edge_keep=edge_keep)
        for layer in self.mlp[:-1]:
            pos_h = ops.linear(pos_h, layer.weight, layer.bias, act=ops.ACT_RELU)
            neg_h = ops.linear(neg_h, layer.weight, layer.bias, act=ops.ACT_RELU)
        pos_h = ops.linear(pos_h, self.mlp[-1].weight, self.mlp[-1].bias)
        neg_h = ops.linear(neg_h, self.mlp[-1].weight, self.mlp[-1].bias)
        return pos_h.sum(1), neg_h.sum(1)

    def _views_ok(self, x, r):
        """bf16 throughput mode with device randomness: both passes share the graph, the weights and -- because the
        corruption is a row permutation, ``x[perm] W^T = (x W^T)[perm]`` -- the first layer's product, so they run
        side by side through every kernel (``GCNEncoder.forward_views``).  Same arithmetic per view."""
        enc = self.encoder
        return (getattr(r, "device_draws", False) and x.is_cuda and x.dim() == 2 and not x.requires_grad
                and hasattr(enc, "forward_views") and x.size(1) % 256 == 0 and enc.views_ok(x.size(0)))

    def _forward_views(self, x, edge_index, edge_keep, r):
        n = x.size(0)
        layer0 = self.encoder.graph_layers[0]
        xw = ops.xw_table(x, layer0.lin.weight)                                  # [N, h] bf16, shared by both views
        perm = r.perm(n, x.device)
        xw2 = ops.pair_rows(xw, perm)                                            # node i: [pos_i | neg_i]
        enc, last = self.encoder, self.mlp[-1]
        conv = enc.graph_layers[-1]
        if (ops.GGD_FOLD_LAST and ops.bf16_storage() and len(self.mlp) == 1 and len(enc.graph_layers) >= 2
                and ops.row_dot_ok(conv.in_channels, torch.bfloat16)):
            # The last conv layer has no activation behind it and the head sums a Linear's outputs, so the scores are
            # LINEAR in that layer's input H:  sum_j (W_m (A_hat (H W_L^T) + b_L) + b_m)_j = A_hat (H w') + b_L . v + sum b_m
            # with v = W_m^T 1 and w' = W_L^T v -- a dot product per row and an aggregation of ONE column per view instead of
            # the layer's [2 N, out] product and [N, 2 out] aggregation (and their backward passes).  Same function of the same
            # parameters, every one of which gets its gradient through autograd of the small products below; the exact-fp32
            # parity mode keeps the layer (reference biomedkg/model/gcl.py:79-90, encoder.py:153-162).
            from ..graph import graph_of
            h2, link = enc.forward_views(xw2, edge_index, 2, edge_keep=edge_keep, before_last=True)   # [N, 2 * hid] bf16
            hid = conv.in_channels
            v = ops.weight_colsum(last.weight)                                      # [out] = W_m^T 1
            w_fold = ops.weight_colsum(conv.lin.weight * v.unsqueeze(1))            # [hid] = W_L^T v  (no BLAS call for 64 k products)
            const = (conv.bias * v).sum() + last.bias.sum()
            u = ops.row_dot(h2.view(2 * n, hid), w_fold, link_in=link).view(n, 2)
            g = graph_of(edge_index, n)
            s = ops.gcn_propagate_narrow(u, g, g.gcn_weights(edge_keep)) + const
            return ops.columns_of_pair(s)
        # [N, 2 * out]; bf16: the embeddings feed the head's product only
        h2 = enc.forward_views(xw2, edge_index, 2, edge_keep=edge_keep, out_bf16=True)
        h = h2.view(2 * n, h2.size(1) // 2)
        for layer in self.mlp[:-1]:
            h = ops.linear(h, layer.weight, layer.bias, act=ops.ACT_RELU)
        last = self.mlp[-1]
        if ops.bf16_storage() and ops.rowsum_linear_ok(h, last.weight):      # (the exact mode keeps the reference's product + sum)
            s = ops.linear_rowsum(h, last.weight, last.bias).view(n, 2)      # (W h + b).sum(1) = h . W.sum(0) + b.sum()
        else:
            s = ops.linear(h, last.weight, last.bias).sum(1).view(n, 2)
        return ops.columns_of_pair(s)
