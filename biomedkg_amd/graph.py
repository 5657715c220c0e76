"""Device-side graph structures behind GCNConv / RGCNConv / the KGE scorer backward.

The reference hands over ``edge_index [2,E] int64`` (row 0 = source, row 1 = target) and
``edge_type [E] int64`` per batch (SURVEY.md 8b batch contract).  Internally the HIP kernels want
destination-sorted int32 CSR (forward) and source-sorted CSR (backward), built once per distinct
edge set and cached on the ``edge_index`` tensor object itself (invalidated by ``_version``).
All index work runs in ``csrc/graph.hip``; this file only owns the buffers.
"""
from __future__ import annotations

import weakref

import torch

from . import _lib as L


def _i32(n, device):
    return torch.empty(max(int(n), 0), dtype=torch.int32, device=device)


class _Sorter:
    """Stable radix sort of 32-bit keys ``lo + hi * n_lo`` carrying the original positions."""

    @staticmethod
    def sort(lo, hi, n_lo, n_hi):
        L.require_cuda(lo, hi)
        n = lo.numel()
        dev = lo.device
        ws_bytes = L.lib().bkg_sort_workspace_bytes(max(n, 1))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        keys_tmp, iota_tmp, sorted_keys, perm = (_i32(n, dev) for _ in range(4))
        L.call("bkg_sort_edges", L.ptr(lo), L.ptr(hi), n_lo, n_hi, n, L.ptr(keys_tmp), L.ptr(iota_tmp),
               L.ptr(sorted_keys), L.ptr(perm), L.ptr(ws), ws_bytes, L.stream())
        return sorted_keys, perm, ws

    @staticmethod
    def presorted_rowptr(lo, hi, n_lo, n_hi, n_queries, stride=1):
        """``sort_rowptr`` for keys that are already ascending (the caller vouches for it): no sort, two launches."""
        L.require_cuda(lo, hi)
        n, dev = lo.numel(), lo.device
        sorted_keys, perm = _i32(n, dev), _i32(n, dev)
        rowptr = _i32(n_queries, dev)
        L.call("bkg_keys_rowptr_presorted", L.ptr(lo), L.ptr(hi), n_lo, n_hi, n, L.ptr(sorted_keys), L.ptr(perm), n_queries,
               stride, L.ptr(rowptr), L.stream())
        return sorted_keys, perm, rowptr

    @staticmethod
    def sort_rowptr(lo, hi, n_lo, n_hi, n_queries, stride=1):
        """``sort`` + the row pointers ``rowptr[q] = first position with key >= q * stride`` in one library call (a key
        set of a mini-batch: one launch)."""
        L.require_cuda(lo, hi)
        n = lo.numel()
        dev = lo.device
        ws_bytes = L.lib().bkg_sort_workspace_bytes(max(n, 1))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        keys_tmp, iota_tmp, sorted_keys, perm = (_i32(n, dev) for _ in range(4))
        rowptr = _i32(n_queries, dev)
        L.call("bkg_sort_edges_rowptr", L.ptr(lo), L.ptr(hi), n_lo, n_hi, n, L.ptr(keys_tmp), L.ptr(iota_tmp),
               L.ptr(sorted_keys), L.ptr(perm), n_queries, stride, L.ptr(rowptr), L.ptr(ws), ws_bytes, L.stream())
        return sorted_keys, perm, rowptr


SPLIT_CHUNK = 256        # entries per work item of a long row (csrc/rowsplit.hip)
SPLIT_MIN_NNZ = 1 << 15  # below this one wave per row is fine and the table is not worth building


TIGHT_GRIDS = __import__("os").environ.get("BKG_TIGHT_GRIDS", "1") != "0"
TIGHT_AFTER = int(__import__("os").environ.get("BKG_TIGHT_AFTER", "3"))      # launches over a table before its host read
_AS_CAPTURED = 0


class as_captured:
    """Build structures the way a captured step does (launch bounds instead of counts, no host read) although nothing is
    being captured: the padded mini-batch loops run their eager steps under it, so that an eager padded step and its
    replayed graph launch the same kernels over the same tables (bit-equal by construction, tests/test_hip_sampled*.py)."""

    def __enter__(self):
        global _AS_CAPTURED
        _AS_CAPTURED += 1

    def __exit__(self, *exc):
        global _AS_CAPTURED
        _AS_CAPTURED -= 1
        return False


def capturing():
    return _AS_CAPTURED > 0 or torch.cuda.is_current_stream_capturing()


VIEW_GRID_MARK = 16


class RowSplit:
    """Device work-item table cutting rows longer than ``SPLIT_CHUNK`` entries (``bkg_rowsplit``)."""

    def __init__(self, rowptr, n_rows, nnz, chunk=SPLIT_CHUNK, pack=False, col=None, ephemeral=False, superset=None):
        """``pack``: rows of <= 8 entries leave the item list for ``short_desc`` (csrc/rowsplit.hip) -- the tables
        ``bkg_spmm_csr_f32`` walks finish them several per wave; the KGE kernels' tables stay one row per item.
        ``ephemeral``: a table that lives for ONE step however many launches walk it (a ``dropout_edge`` view's CSR,
        ``KeptCSR``): it keeps its array-size bounds -- ``nnz`` may itself be a bound -- and never reads its counts.
        ``superset``: the table of a CSR over the same rows whose every row is AT LEAST as long (the graph a view's kept
        entries come from), same ``chunk`` / ``pack``: a row here cuts into no more items, slots or long rows than there,
        and a short row there is short here -- once that table knows its counts they bound this one's launches too."""
        dev = rowptr.device
        self.ephemeral = bool(ephemeral)
        self._src = (rowptr, n_rows, nnz, chunk)
        self._plain, self._plain_asked = None, 0
        self.pack = bool(pack) and col is not None
        self.n_items_max = n_rows + nnz // chunk
        self.n_slots_max = 2 * (nnz // chunk) + 2
        self.n_long_max = nnz // chunk + 1
        self.n_short_max = n_rows if self.pack else 0
        cnt, scan = _i32(4 * (n_rows + 1), dev), _i32(4 * (n_rows + 1), dev)
        self.bufs = [_i32(self.n_items_max, dev) for _ in range(4)] + [_i32(self.n_long_max, dev) for _ in range(3)]
        self.short_desc = _i32(4 * self.n_short_max, dev) if self.pack else None
        self.short_col = _i32(8 * self.n_short_max, dev) if self.pack else None
        self.totals = torch.zeros(4, dtype=torch.int32, device=dev)
        ws_bytes = L.lib().bkg_sort_workspace_bytes(max(n_rows + 1, 1))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        L.call("bkg_rowsplit_build", L.ptr(rowptr), n_rows, nnz, chunk, int(self.pack), L.ptr(cnt), L.ptr(scan),
               *[L.ptr(b) for b in self.bufs], L.ptr(self.short_desc), L.ptr(col) if self.pack else None,
               L.ptr(self.short_col), L.ptr(self.totals), L.ptr(ws), ws_bytes, L.stream())
        self.c = L.RowSplitStruct(*[b.data_ptr() for b in self.bufs], L.ptr(self.short_desc), L.ptr(self.short_col),
                                  self.totals.data_ptr(),
                                  self.n_items_max, self.n_slots_max, self.n_long_max, self.n_short_max)

        self.exact, self._uses = False, 0
        if superset is not None and superset.exact and superset.pack == self.pack and superset._src[3] == chunk:
            # (+ VIEW_GRID_MARK items: four empty workgroups more than the graph's own launches, so that a counter pass --
            # rocprofv3 rows carry kernel name and grid, nothing else -- can tell a view's launches from the graph's)
            self.n_items_max = self.c.n_items_max = min(self.n_items_max, superset.n_items_max + VIEW_GRID_MARK)
            self.n_slots_max = self.c.n_slots_max = min(self.n_slots_max, superset.n_slots_max)
            self.n_long_max = self.c.n_long_max = min(self.n_long_max, superset.n_long_max)

    def use(self):
        """Called by every launch over this table.  From the 3rd launch on the bounds are the table's counts
        (``tighten``: one host read) -- a structure that lives for one or two launches (mini-batches; the bench's
        ``ms_per_step_fresh_structures``) keeps its array-size bounds and never makes the host wait."""
        self._uses += 1
        if self.ephemeral:
            return self
        if not self.exact and self._uses >= TIGHT_AFTER and TIGHT_GRIDS and not capturing():
            self.tighten()
        return self

    def tighten(self):
        """Replace the launch bounds (array-size upper bounds: no host read needed to build the table) by the table's
        actual counts -- ONE host read per table, i.e. per graph structure, never per step.  On the bench graph's
        (relation, target) segments the bounds are 288 k items for 57 k real ones; the 58 k dead blocks cost ~20 us of
        each 190 us launch.  Skipped while a stream is being captured (the captured mini-batch steps keep the bounds)."""
        n_items, n_slots, n_long, n_short = self.totals.tolist()
        self.n_items_max, self.n_long_max, self.n_short_max = n_items, n_long, n_short
        self.n_slots_max = max(n_slots, 1)
        self.c.n_items_max, self.c.n_slots_max, self.c.n_long_max, self.c.n_short_max = n_items, self.n_slots_max, n_long, n_short
        self.exact = True

    def plain(self):
        """The same rows as one item each (no short-row table): rows of 64 sixteen-byte pieces or more already fill a
        wave, and finishing the short ones in the second launch costs more than it saves there (two-view GCN propagate,
        d = 512 bf16: 331 vs 335 us per launch).  Built when the table is asked for the 5th time -- a structure that lives
        for one step (mini-batches, ``ms_per_step_fresh_structures``) never pays for a second table --, kept with this one."""
        if not self.pack or not PLAIN_WIDE_ROWS:
            return self
        self._plain_asked += 1
        if self._plain is None and self._plain_asked < 5:
            return self
        if self._plain is None:
            rowptr, n_rows, nnz, chunk = self._src
            self._plain = RowSplit(rowptr, n_rows, nnz, chunk=chunk, pack=False)
        return self._plain

    def partials(self, d, device):
        return torch.empty(self.n_slots_max, d, dtype=torch.float32, device=device)


PACK_SHORT_ROWS = __import__("os").environ.get("BKG_PACK_SHORT_ROWS", "1") != "0"
PLAIN_WIDE_ROWS = __import__("os").environ.get("BKG_PLAIN_WIDE_ROWS", "1") != "0"


def rowsplit_for(rowptr, n_rows, nnz, pack=False, col=None):
    return RowSplit(rowptr, n_rows, nnz, pack=pack and PACK_SHORT_ROWS, col=col) if nnz >= SPLIT_MIN_NNZ else None


class CSR:
    """``rowptr [n_rows+1]``, ``col [nnz]`` (the other endpoint), ``perm [nnz]`` (original position)."""

    _split = False

    @property
    def split(self):
        if self._split is False:
            self._split = rowsplit_for(self.rowptr, self.n_rows, self.nnz, pack=True, col=self.col)
        return self._split

    def __init__(self, key, other, n_rows, presorted=False, shared=None):
        """``presorted``: ``key`` is already ascending (rows in order, entries of a row in edge order).
        ``shared = (tensor, slot)``: the sort of ``key`` is kept on that tensor under that slot and taken from there when
        it is already known -- the GCN structure's by-source CSR and the R-GCN structure's source side sort the same row of
        the same ``edge_index`` (a training run over both modules, the bench: one sort instead of two)."""
        key = key.contiguous()
        other = other.contiguous()
        dev = key.device
        self.n_rows, self.nnz = int(n_rows), key.numel()
        build = _Sorter.presorted_rowptr if presorted else _Sorter.sort_rowptr
        if shared is not None and shared[0] is not None:
            self.sorted_keys, self.perm, self.rowptr = _cached(shared[0], shared[1], (self.n_rows,),
                                                               lambda: build(key, None, n_rows, 1, n_rows + 1))
        else:
            self.sorted_keys, self.perm, self.rowptr = build(key, None, n_rows, 1, n_rows + 1)
        self.col = _i32(self.nnz, dev)
        L.call("bkg_gather_i64_to_i32", L.ptr(other), L.ptr(self.perm), L.ptr(self.col), self.nnz, L.stream())

    def permute_f32(self, v_edge):
        """Values given in original edge order -> this CSR's order."""
        out = torch.empty(self.nnz, dtype=torch.float32, device=v_edge.device)
        L.call("bkg_gather_f32", L.ptr(v_edge), L.ptr(self.perm), L.ptr(out), self.nnz, L.stream())
        return out


class KeptCSR:
    """One of the two CSRs of a ``dropout_edge`` view (``Graph.gcn_view`` / ``bkg_gcn_view``): the graph's entries the view
    keeps, with their gcn_norm weights -- reference biomedkg/model/gcl.py:42-43,76: PyG hands GCNConv
    ``edge_index[:, keep]``, the dropped edges' rows are never gathered.  The arrays keep the graph's length (``count``,
    on the device, says how many entries are live: nothing here waits for the host); the work-item table keeps its bounds
    (``RowSplit`` ephemeral, bounded by the graph's own table once that one knows its counts)."""

    __slots__ = ("rowptr", "col", "w", "count", "n_rows", "nnz", "split")

    def __init__(self, rowptr, col, w, count, csr):
        self.rowptr, self.col, self.w, self.count = rowptr, col, w, count
        self.n_rows, self.nnz = csr.n_rows, csr.nnz
        self.split = (RowSplit(rowptr, csr.n_rows, csr.nnz, pack=PACK_SHORT_ROWS, col=col, ephemeral=True, superset=csr.split)
                      if csr.nnz >= SPLIT_MIN_NNZ else None)

    @classmethod
    def of_weights(cls, csr, w):
        """The entries of ``csr`` with a non-zero weight ``w`` (``bkg_csr_compact``): a view of a CSR that already carries the
        view's weights -- a node-partitioned graph's (table slots as columns, self loops as entries, global degrees)."""
        dev, n, nnz = w.device, csr.n_rows, csr.nnz
        rowptr, col, count = _i32(n + 1, dev), _i32(nnz, dev), _i32(1, dev)
        w_out = torch.empty(nnz, dtype=torch.float32, device=dev)
        flag, offs = _i32(nnz, dev), _i32(nnz, dev)
        ws_bytes = L.lib().bkg_csr_compact_workspace_bytes(nnz)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        L.call("bkg_csr_compact", L.ptr(csr.rowptr), L.ptr(csr.col), L.ptr(w), n, nnz, L.ptr(flag), L.ptr(offs), L.ptr(rowptr),
               L.ptr(col), L.ptr(w_out), L.ptr(count), L.ptr(ws), ws_bytes, L.stream())
        return cls(rowptr, col, w_out, count, csr)


# a dropout_edge view of a graph of this many edges or more propagates over its kept entries only (KeptCSR): two
# CSRs + two work-item tables per mask (~0.2 ms on the bench graph, less than the masked form's gcn_norm + two weight
# permutations) against 40 % of the gathered rows of the eight
# propagates of a step.  Mini-batch graphs stay on the masked weights (launch-bound: the extra launches cost more).
PLANT_RELATION_ORDER = __import__("os").environ.get("BKG_PLANT_RELATION_ORDER", "1") != "0"
KEPT_CSR_MIN_EDGES = int(__import__("os").environ.get("BKG_KEPT_CSR_MIN_EDGES", str(1 << 20)))


class GcnWeights:
    """gcn_norm result for one (graph, keep mask): weights in both CSR orders + self-loop weight.  ``dst`` / ``src``: the
    view's own CSRs (``KeptCSR``) when the mask was compacted -- ``w_dst`` / ``w_src`` are then THEIR weights."""

    __slots__ = ("w_dst", "w_src", "self_w", "dst", "src")

    def __init__(self, w_dst, w_src, self_w, dst=None, src=None):
        self.w_dst, self.w_src, self.self_w = w_dst, w_src, self_w
        self.dst, self.src = dst, src


class _EdgeHolder:
    """How a structure refers to the ``edge_index`` it was built from.  A structure cached ON that tensor (``graph_of`` /
    ``rel_graph_of``) must not point back at it strongly: tensor -> cache -> structure -> tensor is a cycle the collector
    cannot always break (row views keep the base alive on the C++ side), and every mini-batch of a sampled run would leave
    its structures behind.  ``_weaken`` turns the reference weak once the structure is known to live and die with the
    tensor; a structure built by hand keeps its strong reference."""

    def _hold(self, edge_index):
        self._ei_strong, self._ei_weak = edge_index, None

    def _weaken(self, owner):
        if self._ei_strong is owner:
            self._ei_strong, self._ei_weak = None, weakref.ref(owner)

    @property
    def edge_index(self):
        return self._ei_strong if self._ei_strong is not None else self._ei_weak()

    @property
    def src(self):
        return self.edge_index[0]

    @property
    def dst(self):
        return self.edge_index[1]


class Graph(_EdgeHolder):
    """Homogeneous message-passing graph (GCNConv).  ``num_nodes`` = rows of x."""

    def __init__(self, edge_index, num_nodes):
        L.require_cuda(edge_index)
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError("edge_index must be int64 [2, E]")
        # a device NeighborLoader batch lists its edges target by target in local-id order and says so on the tensor
        # (samplers.NeighborSampler, sampled_gcl): the target-side CSR then needs no sort
        dst_sorted = getattr(edge_index, "_bkg_dst_sorted", None) == edge_index._version      # (vouched for THIS version of the tensor)
        self._hold(edge_index.contiguous())
        self.num_nodes, self.num_edges = int(num_nodes), edge_index.size(1)
        self.by_dst = CSR(self.dst, self.src, num_nodes, presorted=dst_sorted)   # rows = targets, col = sources  (forward)
        self.by_src = CSR(self.src, self.dst, num_nodes, shared=(self.edge_index, "sort_by_source"))   # rows = sources, col = targets  (backward)
        self._plain = None
        self._kept = None

    def gcn_norm_raw(self, keep=None):
        """PyG gcn_norm over this graph (optionally a ``dropout_edge`` view): ``(w_edge [E] in edge order, self_w [N])``
        -- existing self loops get weight 0, every node the unit loop ``self_w = 1 / deg``."""
        dev = self.edge_index.device
        if keep is not None:
            keep = keep.to(torch.uint8).contiguous()
        dis = torch.empty(self.num_nodes, dtype=torch.float32, device=dev)
        self_w = torch.empty_like(dis)
        w_edge = torch.empty(self.num_edges, dtype=torch.float32, device=dev)
        c = self.by_dst
        L.call("bkg_gcn_norm", L.ptr(c.rowptr), L.ptr(c.col), L.ptr(c.perm), L.ptr(self.src), L.ptr(self.dst),
               L.ptr(keep), self.num_nodes, self.num_edges, L.ptr(dis), L.ptr(self_w), L.ptr(w_edge), L.stream())
        return w_edge, self_w

    def gcn_view(self, keep) -> GcnWeights:
        """The ``dropout_edge`` view ``keep`` (uint8 / bool ``[E]``, edge order) as CSRs of its own with their gcn_norm
        weights (``bkg_gcn_view``: flags, one scan, row pointers + degrees, scatter -- five launches, no host read)."""
        dev = self.edge_index.device
        n, e = self.num_nodes, self.num_edges
        keep = keep.to(torch.uint8).contiguous()
        d, s = self.by_dst, self.by_src
        flag, offs = _i32(2 * e, dev), _i32(2 * e, dev)
        dis = torch.empty(n, dtype=torch.float32, device=dev)
        self_w = torch.empty_like(dis)
        rp_d, rp_s, col_d, col_s = _i32(n + 1, dev), _i32(n + 1, dev), _i32(e, dev), _i32(e, dev)
        w_d, w_s = (torch.empty(e, dtype=torch.float32, device=dev) for _ in range(2))
        counts = _i32(2, dev)
        ws_bytes = L.lib().bkg_gcn_view_workspace_bytes(e)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        L.call("bkg_gcn_view", L.ptr(d.rowptr), L.ptr(d.col), L.ptr(d.perm), L.ptr(d.sorted_keys), L.ptr(s.rowptr), L.ptr(s.col),
               L.ptr(s.perm), L.ptr(s.sorted_keys), L.ptr(keep), n, e, L.ptr(flag), L.ptr(offs), L.ptr(dis), L.ptr(self_w),
               L.ptr(rp_d), L.ptr(col_d), L.ptr(w_d), L.ptr(rp_s), L.ptr(col_s), L.ptr(w_s), L.ptr(counts), L.ptr(ws), ws_bytes,
               L.stream())
        kd, ks = KeptCSR(rp_d, col_d, w_d, counts[0:1], d), KeptCSR(rp_s, col_s, w_s, counts[1:2], s)
        return GcnWeights(w_d, w_s, self_w, dst=kd, src=ks)

    def gcn_weights(self, keep=None) -> GcnWeights:
        """``keep``: optional uint8/bool ``[E]`` dropout_edge mask in original edge order."""
        if keep is None and self._plain is not None:
            return self._plain
        if keep is not None and self._kept is not None and self._kept[0] is keep and self._kept[1] == keep._version:
            return self._kept[2]      # every layer (and both GGD views) of a step passes the same mask tensor
        if keep is not None and self.num_edges >= KEPT_CSR_MIN_EDGES:
            out = self.gcn_view(keep)
        else:
            w_edge, self_w = self.gcn_norm_raw(keep)
            out = GcnWeights(self.by_dst.permute_f32(w_edge), self.by_src.permute_f32(w_edge), self_w)
        if keep is None:
            self._plain = out
        else:
            self._kept = (keep, keep._version, out)
        return out


class RelGraph(_EdgeHolder):
    """Relational graph (RGCNConv, aggr="mean"): edges grouped into (relation, target) segments.

    Forward:  H[s] = mean_{e in s} x[src_e]  (spmm over ``seg_ptr``) -> per-relation GEMM -> sum of the
    segment rows of each target (spmm over ``dst_csr``).  Backward uses ``src_csr`` whose entries
    point at segments with weight 1/|segment|.
    """

    def __init__(self, edge_index, edge_type, num_nodes, num_relations, n_src=None, static_segments=False,
                 defer_source_side=False, edge_weight=None):
        """``n_src``: number of SOURCE rows when they live in another index space than the ``num_nodes`` targets (a
        node-partitioned graph: sources are slots of the all-gathered table, targets the rank's own nodes).
        ``static_segments`` (padded mini-batches of a captured step, biomedkg_amd/sampled.py): no host read of the
        segment count -- the structure always carries ``E`` segments, the ones past the real count are EMPTY rows of
        the last relation, their targets spread over the nodes: zero rows in, zeros added.
        ``edge_weight`` (fp32, edge order): per-edge weights INSTEAD of the segment means' ``1 / |segment|`` -- the
        backward side of a node-partitioned layer (biomedkg_amd/partition.py) regroups the forward's weighted edges by
        (relation, source)."""
        L.require_cuda(edge_index, edge_type, edge_weight)
        self._hold(edge_index.contiguous())
        self.edge_type = edge_type.contiguous()
        dev = edge_index.device
        self.num_nodes, self.num_edges, self.num_relations = int(num_nodes), edge_index.size(1), int(num_relations)
        n, e, r = self.num_nodes, self.num_edges, self.num_relations
        src, dst = self.src, self.dst
        sorted_keys, perm, ws = _Sorter.sort(dst, self.edge_type, n, r)
        self.perm = perm
        if PLANT_RELATION_ORDER and e and num_nodes * (r + 1) < (1 << 32):
            # this sort IS a grouping of the triplets by relation (keys = relation * n + target): the scorer's relation
            # order (ops.relation_order_of, cached on the id tensor) needs no sort of its own while the structure is young
            # (a mini-batch, a graph rebuilt every step); a graph that stays gets the stable order on its third step --
            # triplet ids ascending inside a relation: the scorer's id reads and record writes then run forward in memory
            # (the full-graph positives: 0.519 against 0.527 ms per launch)
            rel_edge_ptr = _i32(r + 1, dev)
            L.call("bkg_lower_bound_u32", L.ptr(sorted_keys), e, r + 1, n, L.ptr(rel_edge_ptr), L.stream())
            _cached(self.edge_type, "rel_order_planted", (r,), lambda: [perm, rel_edge_ptr, 0])
        self.src_sorted = _i32(e, dev)
        L.call("bkg_gather_i64_to_i32", L.ptr(src), L.ptr(perm), L.ptr(self.src_sorted), e, L.stream())
        flag, scan = _i32(e, dev), _i32(e, dev)
        self.seg_of = _i32(e, dev)              # sorted edge -> segment
        seg_ptr, seg_dst, seg_rel = _i32(e + 1, dev), _i32(e, dev), _i32(e, dev)
        self.inv_cnt = torch.empty(e, dtype=torch.float32, device=dev)   # per sorted edge
        n_seg = torch.empty(1, dtype=torch.int32, device=dev)
        if static_segments:      # surplus segments: empty, last relation, targets spread over the nodes (they add zeros)
            L.call("bkg_rgcn_segments_static", L.ptr(sorted_keys), e, n, r - 1, L.ptr(flag), L.ptr(scan), L.ptr(self.seg_of),
                   L.ptr(seg_ptr), L.ptr(seg_dst), L.ptr(seg_rel), L.ptr(self.inv_cnt), L.ptr(n_seg), L.ptr(ws),
                   ws.numel(), L.stream())
        else:
            L.call("bkg_rgcn_segments", L.ptr(sorted_keys), e, n, L.ptr(flag), L.ptr(scan), L.ptr(self.seg_of),
                   L.ptr(seg_ptr), L.ptr(seg_dst), L.ptr(seg_rel), L.ptr(self.inv_cnt), L.ptr(n_seg), L.ptr(ws),
                   ws.numel(), L.stream())
        if edge_weight is not None:
            L.call("bkg_gather_f32", L.ptr(edge_weight), L.ptr(perm), L.ptr(self.inv_cnt), e, L.stream())
        self.num_segments = e if static_segments else int(n_seg.item())   # one host sync per distinct graph (sizes H)
        s = self.num_segments
        self.seg_ptr, self.seg_dst, self.seg_rel = seg_ptr[: s + 1], seg_dst[:s], seg_rel[:s]
        # segment ranges per relation (segments are sorted by (rel, dst))
        self.rel_ptr = _i32(r + 1, dev)
        seg_rel_u = self.seg_rel.contiguous()
        L.call("bkg_lower_bound_u32", L.ptr(seg_rel_u), s, r + 1, 1, L.ptr(self.rel_ptr), L.stream())
        # target -> its segments
        self.dst_csr = CSR(self.seg_dst.to(torch.int64), _iota64(s, dev), n)
        self.seg_iota_ptr = iota(s + 1, torch.int32, dev)            # 0 .. s
        self.seg_split = rowsplit_for(self.seg_ptr, s, e, pack=True, col=self.src_sorted)
        self._n_src = n if n_src is None else int(n_src)
        if not defer_source_side:
            self.build_source_side()

    def build_source_side(self):
        """source -> (segment, 1/|segment|) of each out-edge: what the BACKWARD gathers through (the forward never
        reads it, so a mini-batch step builds it on a parallel branch)."""
        e, dev = self.num_edges, self.edge_index.device
        seg_of_edge = torch.empty(e, dtype=torch.int64, device=dev)
        w_edge = torch.empty(e, dtype=torch.float32, device=dev)
        L.call("bkg_rgcn_edge_order", L.ptr(self.perm), L.ptr(self.seg_of), L.ptr(self.inv_cnt), e, L.ptr(seg_of_edge),
               L.ptr(w_edge), L.stream())
        self.src_csr = CSR(self.edge_index[0], seg_of_edge, self._n_src, shared=(self.edge_index, "sort_by_source"))
        self.src_w = self.src_csr.permute_f32(w_edge)


_IOTAS = {}        # (dtype, device) -> [current arange, retired ones]: ONE growing table per kind, handed out as prefixes


def iota(count, dtype, device):
    """``arange(count)`` as a prefix view of a cached table that doubles when outgrown -- the structures of every
    mini-batch ask for a different length, so a table per length would grow without bound.  Outgrown tables stay
    allocated: captured graphs may hold their address."""
    key = (dtype, str(device))
    slot = _IOTAS.setdefault(key, [None, []])
    if slot[0] is None or slot[0].numel() < count:
        if slot[0] is not None:
            slot[1].append(slot[0])
        slot[0] = torch.arange(max(1024, 1 << int(count - 1).bit_length()), dtype=dtype, device=device)
    return slot[0][:count]


def _iota64(count, device):
    return iota(count, torch.int64, device)


_PATTERNS = {}     # (count, modulus, device) -> i % modulus as int32: only the padded mini-batch steps ask (few sizes)


def _spread_pattern(count, modulus, device):
    key = (count, modulus, str(device))
    if key not in _PATTERNS:
        _PATTERNS[key] = (torch.arange(count, device=device) % modulus).to(torch.int32)
    return _PATTERNS[key]


def _cached(t, slot, key, build):
    """One cached object per slot on the tensor object; rebuilt when key or ``_version`` changes."""
    tag = getattr(t, "_bkg_cache", None)
    if tag is None:
        tag = {}
        t._bkg_cache = tag
    key = (t._version,) + key
    hit = tag.get(slot)
    if hit is None or hit[0] != key:
        hit = (key, build())
        tag[slot] = hit
    return hit[1]


def _owned(structure, owner):
    structure._weaken(owner)
    return structure


def graph_of(edge_index, num_nodes) -> Graph:
    return _cached(edge_index, "gcn", (int(num_nodes),), lambda: _owned(Graph(edge_index, num_nodes), edge_index))


def rel_graph_of(edge_index, edge_type, num_nodes, num_relations) -> RelGraph:
    key = (int(num_nodes), int(num_relations), edge_type.data_ptr(), edge_type._version)
    return _cached(edge_index, "rgcn", key,
                   lambda: _owned(RelGraph(edge_index, edge_type, num_nodes, num_relations), edge_index))


def plant_rel_graph(rg: RelGraph):
    """Make ``rel_graph_of(rg.edge_index, rg.edge_type, ...)`` return this (already built) structure."""
    ei, et = rg.edge_index, rg.edge_type
    key = (int(rg.num_nodes), int(rg.num_relations), et.data_ptr(), et._version)
    _cached(ei, "rgcn", key, lambda: _owned(rg, ei))
    return rg
