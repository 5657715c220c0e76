"""GNN encoders behind the reference's class names (``biomedkg/model/encoder.py``).

``GCNConv`` / ``RGCNConv`` keep PyG's parameter names and initialisers so reference checkpoints
load (``graph_layers.{i}.lin.weight`` / ``.bias``; ``.weight [R,in,out]`` / ``.root`` / ``.bias`` --
SURVEY.md 8b).  The layer arithmetic runs in the HIP library: GEMM -> CSR segment-reduce with bias,
ReLU and dropout fused into the epilogue (``biomedkg_amd.ops``).
"""
from __future__ import annotations

import math

import torch

from .. import ops
from .. import rand as _rand
from ..graph import graph_of, rel_graph_of


def _glorot(t):
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


class GCNConv(torch.nn.Module):
    """PyG GCNConv defaults (add_self_loops, normalize, bias)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = torch.nn.Linear(in_channels, out_channels, bias=False)
        self.bias = torch.nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        _glorot(self.lin.weight)
        torch.nn.init.zeros_(self.bias)

    def forward(self, x, edge_index, edge_keep=None, relu=False, drop=None, hidden=False):
        g = graph_of(edge_index, x.size(0))
        return ops.gcn_conv(x, self.lin.weight, self.bias, g, g.gcn_weights(edge_keep), relu=relu, drop=drop,
                            hidden=hidden)


class RGCNConv(torch.nn.Module):
    """PyG RGCNConv defaults (aggr="mean", root_weight, bias, no bases/blocks; Appendix A.3)."""

    def __init__(self, in_channels: int, out_channels: int, num_relations: int):
        super().__init__()
        self.in_channels, self.out_channels, self.num_relations = in_channels, out_channels, num_relations
        self.weight = torch.nn.Parameter(torch.empty(num_relations, in_channels, out_channels))
        self.root = torch.nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = torch.nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        _glorot(self.weight)
        _glorot(self.root)
        torch.nn.init.zeros_(self.bias)

    links = True      # takes ops.GateLink hand-overs from _Stack._run

    def forward(self, x, edge_index, edge_type, relu=False, drop=None, hidden=False, link_in=None, link_out=None):
        rg = rel_graph_of(edge_index, edge_type, x.size(0), self.num_relations)
        return ops.rgcn_conv(x, self.weight, self.root, self.bias, rg, relu=relu, drop=drop, hidden=hidden,
                             link_in=link_in, link_out=link_out)


class _Stack(torch.nn.Module):
    """in->h, L x (h->h), h->out with ReLU + dropout(0.2) after all but the last layer
    (reference encoder.py:21-59 / :138-162); the activation pair is fused into each layer."""

    def __init__(self, make, in_dim, hidden_dim, out_dim, num_hidden_layers, drop_out=True):
        super().__init__()
        self.drop_out = drop_out
        dims = [in_dim] + [hidden_dim] * (num_hidden_layers + 1) + [out_dim]
        self.graph_layers = torch.nn.ModuleList(make(a, b) for a, b in zip(dims[:-1], dims[1:]))
        self.reset_parameters()

    def reset_parameters(self):
        for layer in self.graph_layers:
            layer.reset_parameters()

    def _run(self, x, *graph, **kw):
        if ops.features_bf16_ok(x) and x.is_cuda:
            x = ops.cached_bf16(x)      # constant input features: one bf16 table per tensor, not one cast per step
        last = len(self.graph_layers) - 1
        below = None      # ops.GateLink with the layer below: the layer above may apply that layer's backward gate itself
        for i, layer in enumerate(self.graph_layers):
            lk = {}
            if getattr(layer, "links", False) and ops.bf16_storage():
                above = ops.GateLink() if i != last else None
                lk = {"link_in": below, "link_out": above}
                below = above
            if i == last:
                x = layer(x, *graph, **lk, **kw)
            else:
                drop = None
                if self.drop_out and self.training:
                    drop = _rand.current().dropout((x.size(0), layer.out_channels), 0.2, x.device)
                # hidden=True: in bf16 mode the activation between two conv layers is kept as a bf16 table (ops.py)
                x = layer(x, *graph, relu=True, drop=drop, hidden=True, **lk, **kw)
        return x


class GCNEncoder(_Stack):
    def __init__(self, in_dim: int, hidden_dim: int, out_dim: int, num_hidden_layers: int, drop_out: bool = True):
        super().__init__(GCNConv, in_dim, hidden_dim, out_dim, num_hidden_layers, drop_out)

    def views_ok(self, n_rows, views=2):
        """Can ``views`` inputs over one graph go through the stack side by side (``forward_views``)?"""
        layers = self.graph_layers
        if len(layers) < 2:
            return False
        if ops.bf16_storage():
            return (all(ops.views_ok(n_rows, l.in_channels, l.out_channels, views) for l in layers[1:])
                    and ops.views_ok(n_rows, 256, layers[0].out_channels, views))
        return all(ops.views_ok_f32(n_rows, l.in_channels, l.out_channels, views) for l in layers)     # (exact mode)

    def forward_views(self, xw2, edge_index, views, edge_keep=None, out_bf16=False, before_last=False):
        """The stack over ``views`` inputs of the SAME graph at once (bf16 mode; GGD's positive + corrupted pass).
        ``xw2 [N, views * h]``: the first layer's products ``x_v W_0^T`` side by side (``ops.xw_table``); returns the
        last layer's outputs ``[N, views * out]`` (fp32; bf16 with ``out_bf16`` when only products consume them), view v
        in columns ``[v * out, (v + 1) * out)``.  ``before_last``: stop in front of the last layer and return its INPUT
        (the bf16 hidden activations) and the ``ops.GateLink`` to the layer that made it -- for a caller that folds the last
        layer into what follows (``GGD._forward_views``)."""
        g = graph_of(edge_index, xw2.size(0))
        gw = g.gcn_weights(edge_keep)
        last = len(self.graph_layers) - 1
        h = xw2
        below = None                 # ops.GateLink with the layer below: whose backward applies that layer's gate
        for i, layer in enumerate(self.graph_layers):
            if before_last and i == last:
                return h, below
            drop = None
            if i != last and self.drop_out and self.training:
                drop = _rand.current().dropout((h.size(0), views * layer.out_channels), 0.2, h.device)
            above = ops.GateLink() if i != last else None
            h = ops.gcn_conv_views(h, None if i == 0 else layer.lin.weight, layer.bias, g, gw, views, relu=(i != last),
                                   drop=drop, hidden=(i != last) or out_bf16, is_table=(i == 0), link_in=below,
                                   link_out=above)
            below = above
        return h

    def forward(self, x, edge_index, edge_keep=None):
        return self._run(x, edge_index, edge_keep=edge_keep)


class RGCN(_Stack):
    def __init__(self, in_dim: int, hidden_dim: int, out_dim: int, num_hidden_layers: int, num_relations: int,
                 drop_out: bool = True):
        super().__init__(lambda a, b: RGCNConv(a, b, num_relations), in_dim, hidden_dim, out_dim, num_hidden_layers,
                         drop_out)

    def forward(self, x, edge_index, edge_type):
        return self._run(x, edge_index, edge_type)


class RGAT(torch.nn.Module):
    """Unsupported: the reference's RGAT cannot be constructed (``encoder.py:79-84`` passes
    ``num_relation=`` to RGATConv, ``factory.py:71-79`` passes ``num_relations=`` to RGAT -- TypeError
    either way).  Out of scope by the hot-path table."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("RGAT is broken in the reference and out of scope here")
