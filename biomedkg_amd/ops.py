"""Host-side wrappers (autograd glue) over the C-ABI kernels.  No CPU path: every function
requires CUDA (HIP) tensors and raises otherwise.  fp32 throughout in this file.
"""
from __future__ import annotations

import weakref

import torch

from . import _lib as L
from .graph import CSR, Graph, GcnWeights, KeptCSR, RelGraph

ACT_NONE, ACT_RELU, ACT_ELU = 0, 1, 2

# Matrix-core operand precision of every GEMM on the path: "fp32" = exact fp32 MFMA (default; the 1e-5
# loss-parity bar of BASELINE config 2 needs it), "bf16" = operands rounded to bf16 while staged into LDS,
# fp32 accumulate and fp32 results in HBM (BASELINE config 3 throughput runs).
_PRECISION = 0
if __import__("os").environ.get("BKG_PRECISION", "fp32") == "bf16":   # e.g. BKG_PRECISION=bf16 python train_kge.py ...
    _PRECISION = 1


def set_matmul_precision(name: str):
    global _PRECISION
    if name not in ("fp32", "bf16"):
        raise ValueError("precision must be 'fp32' or 'bf16'")
    _PRECISION = 0 if name == "fp32" else 1


def get_matmul_precision() -> str:
    return "bf16" if _PRECISION else "fp32"


_DET_WS = None
_DET_RETIRED = []     # outgrown workspaces stay allocated: captured graphs may still hold their address


def set_deterministic(on: bool, workspace_floats: int = 1 << 26):
    """Lightning's ``Trainer(deterministic=True)`` (reference train_gcl.py:66, train_kge.py:46) for the HIP path: the
    split-K weight-gradient products add their K chunks through per-chunk slabs + a fixed-order reduce instead of float
    atomics (``bkg_set_deterministic_workspace``); every other kernel on the path is order-fixed already (segment reduces,
    NT-Xent, column sums, the negative sampler).  Costs one extra pass over the slabs per product."""
    global _DET_WS
    if on:
        if not torch.cuda.is_available():
            raise L.HipLibraryError("deterministic mode needs the GPU workspace")
        dev = torch.device("cuda", torch.cuda.current_device())
        if _DET_WS is None or _DET_WS.numel() < workspace_floats or _DET_WS.device != dev:
            if _DET_WS is not None:
                _DET_RETIRED.append(_DET_WS)
            # revive a retired buffer of this device that is large enough before allocating (on/off toggles: bench, tests)
            hit = next((t for t in _DET_RETIRED if t.device == dev and t.numel() >= workspace_floats), None)
            if hit is not None:
                _DET_RETIRED.remove(hit)
            _DET_WS = hit if hit is not None else torch.empty(workspace_floats, dtype=torch.float32, device=dev)
        L.call("bkg_set_deterministic_workspace", L.ptr(_DET_WS), _DET_WS.numel())
    else:
        L.call("bkg_set_deterministic_workspace", None, 0)
        if _DET_WS is not None:
            _DET_RETIRED.append(_DET_WS)      # a graph captured in deterministic mode still launches with its address
        _DET_WS = None


def is_deterministic() -> bool:
    return _DET_WS is not None


def _det_reserve(floats):
    """Deterministic mode: make sure the slab workspace holds ``floats`` (one [M, N] slab per K chunk of the product
    about to be launched); grows it when a wider / longer product comes along."""
    if _DET_WS is not None and (_DET_WS.numel() < floats or _DET_WS.device.index != torch.cuda.current_device()):
        # (one process drives one GPU -- SURVEY 3.5 -- and the slab users are serialised on the launch stream; a second
        # device in the same process gets a workspace of its own here rather than another device's pointer)
        set_deterministic(True, int(max(floats, _DET_WS.numel())))


def _follow_torch_determinism():
    """Real Lightning's ``Trainer(deterministic=True)`` flips ``torch.use_deterministic_algorithms``: follow it."""
    if _DET_WS is None and torch.are_deterministic_algorithms_enabled():
        set_deterministic(True)


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- raw kernels
_REPLAY_GEN = 0


def note_graph_replay():
    """A captured training step was replayed: its optimizer step changed the weights WITHOUT touching their
    ``_version``, so everything cached per weight version is stale for eager code from here on (the replay itself
    rebuilds what it needs inside the graph)."""
    global _REPLAY_GEN
    _REPLAY_GEN += 1


def _weight_bf16(w, transpose):
    """bf16 ``[.., N, K]`` image of a (batch of) weight matrix for ``bkg_gemm_xw`` -- ``w`` itself when it is stored
    ``[N, K]``, its transpose when stored ``[K, N]``.  One tiny launch per weight version writes BOTH orientations
    (the forward needs one, the backward the other)."""
    cache = getattr(w, "_bkg_w16", None)
    key = weight_image_key(w)
    if cache is None or cache[0] != key:
        w3 = w if w.dim() == 3 else w.unsqueeze(0)
        batch, rows, cols = w3.shape
        # the images keep their addresses across versions of a weight (rewritten in place): the fused optimizer step
        # refreshes them itself (optim.clip_adam_step), and a captured step may read them by address
        if (cache is not None and cache[1].shape == w.shape and cache[1].device == w.device
                and not torch.cuda.is_current_stream_capturing()):
            plain, trans = cache[1], cache[2]
            plain3, trans3 = (plain, trans) if w.dim() == 3 else (plain.unsqueeze(0), trans.unsqueeze(0))
        else:
            plain3 = torch.empty(batch, rows, cols, dtype=torch.bfloat16, device=w.device)
            trans3 = torch.empty(batch, cols, rows, dtype=torch.bfloat16, device=w.device)
            plain, trans = (plain3, trans3) if w.dim() == 3 else (plain3[0], trans3[0])
        L.call("bkg_weight_to_bf16", L.ptr(w3), w3.stride(1), w3.stride(0), rows, cols, batch, L.ptr(plain3), plain3.stride(1),
               plain3.stride(0), L.ptr(trans3), trans3.stride(1), trans3.stride(0), L.stream())
        cache = (key, plain, trans)
        w._bkg_w16 = cache
    return cache[2] if transpose else cache[1]


def weight_image_key(w):
    """What a cached bf16 image of ``w`` is valid for: this storage, this version, no graph replay since."""
    return (w.data_ptr(), w._version, _REPLAY_GEN)


_TALL_ROWS = 256    # fewest rows the streaming product is used for (a heuristic, not a kernel limit; a node-partitioned
                    # rank that owns hub nodes may hold only ~1k rows)


def _xw_ok(a, m, n, k):
    """Shapes / layouts the streaming activation-x-weight kernel (csrc/gemm_xw.hip) takes in bf16 mode."""
    el = a.element_size()
    return (_PRECISION == 1 and (m >= _TALL_ROWS or a.dtype == torch.bfloat16) and n * k <= (1 << 22) and k % 256 == 0 and n % 32 == 0 and a.stride(1) == 1 and
            (a.stride(0) * el) % 16 == 0 and a.data_ptr() % 16 == 0)


def _bf16_operand_ok(rows, n, k):
    """Would a contiguous bf16 ``[rows, k]`` table be accepted as the tall operand of an x-weight product with an
    ``[n, k]`` weight?  (Decides whether a producer may emit its result in bf16 only.)"""
    return _PRECISION == 1 and rows >= _TALL_ROWS and n * k <= (1 << 22) and k % 256 == 0 and n % 32 == 0


def _gemm_xw(a, wt, m, n, k, out, out_dtype, alpha, beta, bias, act, gate, gate_scale, group_ptr=None, groups=0,
             a_rows=None):
    if out is None:
        out = torch.empty(m, n, dtype=out_dtype, device=a.device)
    w2 = wt if wt.dim() == 2 else wt[0]
    L.call("bkg_gemm_xw", m, n, k, alpha, L.ptr(a), int(a.dtype == torch.bfloat16), a.stride(0), L.ptr(a_rows), L.ptr(wt),
           w2.stride(0),
           wt.stride(0) if wt.dim() == 3 else 0, L.ptr(group_ptr), groups, beta, L.ptr(out),
           int(out.dtype == torch.bfloat16), out.stride(0), L.ptr(bias), act, L.ptr(gate),
           gate.stride(0) if gate is not None else 0, gate_scale, L.stream())
    return out


def gemm(a, b, trans_a=False, trans_b=False, out=None, alpha=1.0, beta=0.0, bias=None, act=ACT_NONE, gate=None,
         gate_scale=1.0, out_dtype=torch.float32):
    """``out = epi(alpha * op(a) @ op(b))``; a/b 2-D row-major fp32.  fp32 mode: exact-fp32 MFMA kernel.  bf16 mode:
    bf16 operands / fp32 accumulate; tall ``a`` (fp32 or bf16) times a small weight ``b`` streams through
    ``bkg_gemm_xw`` and may write a bf16 ``out`` (``out_dtype``)."""
    L.require_cuda(a, b, out, bias, gate)
    if trans_a:
        _follow_torch_determinism()
        _det_reserve(2048 * 128 * 128)      # split-K: at most 2048 / tiles chunks of tiles * 128 * 128 floats
    if b.dtype != torch.bfloat16:
        b = _f32c(b)
    m, k = (a.size(1), a.size(0)) if trans_a else (a.size(0), a.size(1))
    k2, n = (b.size(1), b.size(0)) if trans_b else (b.size(0), b.size(1))
    if k != k2:
        raise ValueError(f"gemm shape mismatch: {tuple(a.shape)} x {tuple(b.shape)} (ta={trans_a}, tb={trans_b})")
    if (not trans_a and _xw_ok(a, m, n, k) and (out is None or out.stride(0) % 4 == 0) and
            (gate is None or (gate.stride(0) % 4 == 0 and gate.data_ptr() % 16 == 0))):
        if gate is not None:
            gate = _f32c(gate)
        return _gemm_xw(a, _weight_bf16(b, not trans_b), m, n, k, out, out_dtype, alpha, beta, bias, act, gate, gate_scale)
    plain = bias is None and act == ACT_NONE and gate is None and alpha == 1.0 and beta == 0.0
    if (_PRECISION == 1 and trans_a and not trans_b and plain and out_dtype == torch.float32 and k >= 4096
            and ((m + 63) // 64) * ((n + 63) // 64) <= 256 and a.stride(1) == 1):
        # deep-K weight gradient dW = A^T B: split-K kernel, either operand may be a bf16 table
        if out is None:
            out = torch.empty(m, n, dtype=torch.float32, device=a.device)
        bb = b if b.stride(1) == 1 else b.contiguous()
        L.call("bkg_gemm_tn_bf16", m, n, k, L.ptr(a), int(a.dtype == torch.bfloat16), a.stride(0), L.ptr(bb),
               int(bb.dtype == torch.bfloat16), bb.stride(0), L.ptr(out), out.stride(0), L.stream())
        return out
    if trans_a and out_dtype == torch.float32 and (a.dtype == torch.bfloat16 or b.dtype == torch.bfloat16):
        # a weight gradient over few rows (a node-partitioned rank that owns hub nodes holds ~4k rows or fewer: below the
        # split-K kernel's floor): widen the bf16 tables once and take the general kernel
        a, b = a.float(), b.float()
    if out_dtype != torch.float32 or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise TypeError("bf16 activations are only taken / produced by the bf16-mode tall products")
    a = _f32c(a)
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    L.call("bkg_gemm_f32", int(trans_a), int(trans_b), m, n, k, alpha, L.ptr(a), a.stride(0), L.ptr(b), b.stride(0),
           beta, L.ptr(out), out.stride(0), L.ptr(bias), act, L.ptr(gate), gate.stride(0) if gate is not None else 0,
           gate_scale, _PRECISION, L.stream())
    return out


class GateLink:
    """Hand-over between two consecutive conv layers of a stack (bf16 mode).  The layer ABOVE, whose backward makes the
    gradient of the activation between them, may apply the ReLU / dropout backward gate of the layer BELOW in the
    epilogue of the product that makes that gradient (``gemm_gated``) and leave the lower layer's bias gradient here;
    the lower layer's backward then finds ``applied`` set and skips its own gate pass over the table.  The lower
    layer's forward fills ``relu`` / ``scale``; a fresh link per forward pass; the activation must have no other
    reader than the layer above: the links are made and held by ``_Stack._run`` / ``forward_views`` only, which hand a
    hidden activation to nothing but the next layer (a second reader's gradient contribution would arrive un-gated).
    ``applied`` is per backward pass: the upper layer sets it in the backward in which it gated, the lower layer clears it
    when it consumes the hand-over -- a later backward over the same graph in which the upper layer takes its plain
    branch finds it clear and the lower layer gates for itself."""
    __slots__ = ("relu", "scale", "applied", "db")

    def __init__(self):
        self.relu, self.scale, self.applied, self.db = False, 1.0, False, None


GATE_IN_PRODUCT = __import__("os").environ.get("BKG_GATE_IN_PRODUCT", "1") != "0"


def gemm_gated_ok(a, b, gate):
    """Shapes ``gemm_gated`` takes: tall contiguous bf16 ``a [m, 256]`` and ``gate [m, 256]``, fp32 ``b [256, 256]``."""
    return (GATE_IN_PRODUCT and bf16_storage() and a.dtype == torch.bfloat16 and gate.dtype == torch.bfloat16
            and a.dim() == 2 and gate.dim() == 2 and a.size(0) >= _TALL_ROWS and a.size(1) == 256 and tuple(b.shape) == (256, 256)
            and gate.shape == a.shape and a.is_contiguous() and gate.is_contiguous()
            and a.data_ptr() % 16 == 0 and gate.data_ptr() % 16 == 0)


def gemm_gated(a, b, gate, gate_scale):
    """``(dx, colsum(dx))`` with ``dx = bf16(gate > 0 ? bf16(a @ b) * gate_scale : 0)`` (csrc/gemm_xw.hip
    ``bkg_gemm_xw8_gated``): the values of ``to_bf16_colsum(gemm(a, b, out_dtype=bf16), gate, gate_scale)`` without the
    pass over the table in between (the column sums add the same numbers in another order)."""
    L.require_cuda(a, b, gate)
    if not gemm_gated_ok(a, b, gate):
        raise ValueError("gemm_gated: tall contiguous bf16 [m, 256] operands and a [256, 256] weight (gemm_gated_ok)")
    m = a.size(0)
    wt = _weight_bf16(_f32c(b) if b.dtype != torch.bfloat16 else b, True)      # [n, k] image of b [k, n]
    out = torch.empty(m, 256, dtype=torch.bfloat16, device=a.device)
    ws = torch.empty(L.lib().bkg_gemm_xw8_gated_workspace_floats(m), dtype=torch.float32, device=a.device)
    cs = torch.empty(256, dtype=torch.float32, device=a.device)
    L.call("bkg_gemm_xw8_gated", m, L.ptr(a), a.stride(0), L.ptr(wt), wt.stride(0), L.ptr(out), out.stride(0), L.ptr(gate),
           gate.stride(0), float(gate_scale), L.ptr(ws), L.ptr(cs), L.stream())
    return out, cs


def gemm_grouped_rows(a, b, group_ptr, trans_b=False, out_dtype=torch.float32, a_rows=None):
    """``out[rows of g] = a[rows of g] @ op(b[g])``; ``b`` is ``[G, k, n]`` (or ``[G, n, k]`` if trans_b).
    ``a_rows`` (int32, bf16 mode's streaming kernel only): output row ``m`` reads ``a[a_rows[m]]``."""
    L.require_cuda(a, b, group_ptr, a_rows)
    b = _f32c(b)
    g = b.size(0)
    k, n = (b.size(2), b.size(1)) if trans_b else (b.size(1), b.size(2))
    rows = a.size(0) if a_rows is None else a_rows.numel()
    if _xw_ok(a, rows, n, k):
        return _gemm_xw(a, _weight_bf16(b, not trans_b), rows, n, k, None, out_dtype, 1.0, 0.0, None, ACT_NONE, None,
                        1.0, group_ptr, g, a_rows)
    if a_rows is not None:
        raise ValueError("gathered rows are only taken by the bf16-mode streaming product")
    if out_dtype != torch.float32:
        raise TypeError("bf16 outputs are only produced by the bf16-mode activation-x-weight products")
    a = _f32c(a)
    out = torch.empty(a.size(0), n, dtype=torch.float32, device=a.device)
    L.call("bkg_gemm_grouped_rows_f32", int(trans_b), a.size(0), n, k, L.ptr(a), a.stride(0), L.ptr(b), b.stride(1),
           b.stride(0), L.ptr(out), out.stride(0), L.ptr(group_ptr), g, None, 0, 1.0, _PRECISION,
           L.stream())
    return out


def gathered_k_ok(a, b):
    """``gemm_grouped_k(.., b_rows=..)``: two bf16 tables whose widths are multiples of the 128-column tile."""
    return (_PRECISION == 1 and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.size(1) % 128 == 0
            and b.size(1) % 128 == 0 and a.stride(1) == 1 and b.stride(1) == 1 and a.stride(0) % 8 == 0
            and b.stride(0) % 8 == 0)


def gemm_grouped_k(a, b, group_ptr, num_groups, b_rows=None):
    """``out[g] = a[rows of g]^T @ b[rows of g]`` -> ``[G, a.cols, b.cols]`` (split-K, f32 atomics).
    ``b_rows`` (int32; see :func:`gathered_k_ok`): row ``k`` of the second operand is ``b[b_rows[k]]``."""
    L.require_cuda(a, b, group_ptr, b_rows)
    _follow_torch_determinism()
    m, n = a.size(1), b.size(1)
    chunk = L.lib().bkg_gemm_grouped_k_chunk(a.size(0), m, n, _PRECISION) if _PRECISION == 0 else 512
    _det_reserve((a.size(0) // chunk + num_groups + 1) * m * n)     # full chunks + one partial chunk per group
    # atomics accumulate into a zeroed result; the deterministic fp32 path's fixed-order reduce writes every element
    new = torch.empty if (_PRECISION == 0 and _DET_WS is not None) else torch.zeros
    out = new(num_groups, m, n, dtype=torch.float32, device=a.device)
    if b_rows is not None and not gathered_k_ok(a, b):
        raise ValueError("gathered rows need two bf16 tables with 128-aligned widths (bf16 mode)")
    if _PRECISION == 1:
        a = a if a.stride(1) == 1 else a.contiguous()
        b = b if b.stride(1) == 1 else b.contiguous()
        L.call("bkg_gemm_grouped_k_bf16", a.size(0), m, n, L.ptr(a), int(a.dtype == torch.bfloat16), a.stride(0), L.ptr(b),
               int(b.dtype == torch.bfloat16), b.stride(0), L.ptr(b_rows), L.ptr(out), n, m * n, L.ptr(group_ptr),
               num_groups, L.stream())
        return out
    a, b = _f32c(a), _f32c(b)
    L.call("bkg_gemm_grouped_k_f32", a.size(0), m, n, L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), L.ptr(out), n,
           m * n, L.ptr(group_ptr), num_groups, _PRECISION, L.stream())
    return out


def colsum(x):
    """``x.sum(0)`` in fp32 for a tall fp32 / bf16 table (bias gradients): one streaming pass (csrc/colsum.hip)."""
    L.require_cuda(x)
    epl = 8 if x.dtype == torch.bfloat16 else 4
    n, d = x.shape
    if (x.dtype not in (torch.float32, torch.bfloat16) or x.stride(1) != 1 or d % epl or d // epl > 256
            or x.stride(0) % epl or x.data_ptr() % 16):
        return x.sum(0, dtype=torch.float32)
    ws = torch.empty(L.lib().bkg_colsum_workspace_floats(n, d), dtype=torch.float32, device=x.device)
    out = torch.empty(d, dtype=torch.float32, device=x.device)
    L.call("bkg_colsum_f32", L.ptr(x), int(x.dtype == torch.bfloat16), x.stride(0), n, d, L.ptr(ws), L.ptr(out), L.stream())
    return out


class Dropout:
    """Dropout request for a kernel epilogue: explicit keep mask (replay) or in-kernel Philox."""

    __slots__ = ("p", "keep", "seed", "stream_id")

    def __init__(self, p, keep=None, seed=0, stream_id=0):
        self.p, self.keep, self.seed, self.stream_id = float(p), keep, int(seed), int(stream_id)


# bench.py sets this to a dict to time tagged launches with HIP events on the launch stream:
# tag -> {"events": [(start, end), ...], "bytes": algorithmic bytes of one launch}
KERNEL_TIMERS = None


_EVENTS = []


def reserve_events(count):
    """Timing events for the tagged calls, created AND recorded once ahead of a timed region: the runtime grows its event /
    signal pools lazily, and on a fresh process that growth showed up as one ~70 ms step inside bench.py's timed loop."""
    fresh = [torch.cuda.Event(enable_timing=True) for _ in range(max(0, count - len(_EVENTS)))]
    for ev in fresh:
        ev.record()
    torch.cuda.synchronize()
    _EVENTS.extend(fresh)


def _event():
    return _EVENTS.pop() if _EVENTS else torch.cuda.Event(enable_timing=True)


class _Timed:
    """HIP events around one tagged library call on the launch stream, when bench.py collects them (KERNEL_TIMERS)."""

    def __init__(self, tag, algorithmic_bytes, **extra):
        self.on = KERNEL_TIMERS is not None and tag is not None
        self.tag, self.bytes, self.extra = tag, algorithmic_bytes, extra

    def __enter__(self):
        if self.on:
            self.ev0, self.ev1 = _event(), _event()
            self.ev0.record()
        return self

    def __exit__(self, *exc):
        if self.on and exc[0] is None:
            self.ev1.record()
            by = self.bytes() if callable(self.bytes) else self.bytes
            rec = KERNEL_TIMERS.setdefault(self.tag, {"events": [], "bytes": by, **self.extra})
            rec["events"].append((self.ev0, self.ev1))
        return False


def _spmm_bytes(nnz, n_rows, d, has_w, elem=4, out_elem=4):
    """Algorithmic bytes of one launch: gathered row (fp32 or bf16 table) + int32 index
    (+ fp32 weight) per entry, stored row (fp32 or bf16) + int32 rowptr per output row."""
    return nnz * (d * elem + 4 + (4 if has_w else 0)) + n_rows * (d * out_elem + 4)


def timer_bytes(rec):
    """Algorithmic bytes per launch of a ``KERNEL_TIMERS`` record; a call over a compacted view (``live``: the device
    count of its entries, the last launch's) is priced on the entries it walks.  Reads the device: after the timed loop."""
    if rec.get("live") is not None:
        rec["bytes"], rec["live"] = _spmm_bytes(int(rec["live"].item()), *rec["bytes_of"]), None
    return rec["bytes"]


def to_bf16(x, gate=None, gate_scale=1.0):
    """bf16 gather table of ``x`` (csrc/spmm.hip ``bkg_gate_cast_bf16``), optionally through the ReLU/dropout
    backward gate ``gate > 0 ? x * gate_scale : 0``.  ``x`` and ``gate`` may each be fp32 or bf16."""
    L.require_cuda(x, gate)
    if x.dtype == torch.bfloat16 and gate is None:
        return x if x.is_contiguous() else x.contiguous()
    x = x if x.is_contiguous() else x.contiguous()
    if gate is not None:
        gate = gate if gate.is_contiguous() else gate.contiguous()
    for t in (x, gate):
        if t is not None and t.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f"expected float32 / bfloat16, got {t.dtype}")
    if x.numel() % 8:
        raise ValueError("to_bf16 needs a multiple of 8 elements")
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    L.call("bkg_gate_cast_bf16", L.ptr(x), int(x.dtype == torch.bfloat16), L.ptr(gate),
           int(gate is not None and gate.dtype == torch.bfloat16), gate_scale, L.ptr(out), x.numel(), L.stream())
    return out


def to_bf16_colsum(x, gate=None, gate_scale=1.0):
    """``(to_bf16(x, gate, gate_scale), its column sums)`` in ONE pass over a tall contiguous table -- the gated output
    gradient of a layer and its bias gradient (bit-identical to ``colsum`` over the finished table).  The sums are
    ``None`` where no cast runs at all (``x`` already bf16, no gate) or the shape is out of the kernel's range: the
    caller then sums the table itself."""
    L.require_cuda(x, gate)
    if x.dtype == torch.bfloat16 and gate is None:
        return (x if x.is_contiguous() else x.contiguous()), None
    n, d = (x.size(0), x.size(1)) if x.dim() == 2 else (0, 0)
    if (n <= 4096 or d % 8 or d // 8 > 256 or not x.is_contiguous() or (gate is not None and not gate.is_contiguous())
            or x.dtype not in (torch.float32, torch.bfloat16) or (gate is not None and gate.dtype not in (torch.float32, torch.bfloat16))):
        return to_bf16(x, gate, gate_scale), None
    out = torch.empty(n, d, dtype=torch.bfloat16, device=x.device)
    ws = torch.empty(L.lib().bkg_colsum_workspace_floats(n, d), dtype=torch.float32, device=x.device)
    cs = torch.empty(d, dtype=torch.float32, device=x.device)
    L.call("bkg_gate_cast_colsum_bf16", L.ptr(x), int(x.dtype == torch.bfloat16), L.ptr(gate),
           int(gate is not None and gate.dtype == torch.bfloat16), gate_scale, L.ptr(out), n, d, L.ptr(ws), L.ptr(cs), L.stream())
    return out, cs


def masked_bf16(x, p, seed, stream_id):
    """``bf16(x * (rand >= p))`` -- the GCL feature-masking augmentation fused with the bf16 table cast."""
    L.require_cuda(x)
    x = _f32c(x)
    src = x
    if features_bf16_ok(x):          # constant features: the mask is applied to their cached bf16 copy (the same values:
        src = cached_bf16(x)         # rounding and zeroing commute), 200 MB read instead of 400 at the bench size
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    L.call("bkg_mask_cast_bf16", L.ptr(src), int(src.dtype == torch.bfloat16), float(p), seed, stream_id, L.ptr(out), x.numel(),
           L.stream())
    return out


def features_bf16_ok(x):
    """An fp32 feature matrix that bf16 mode may hand to the first conv layer as a bf16 table."""
    return (bf16_storage() and x.dtype == torch.float32 and x.dim() == 2 and not x.requires_grad and x.size(0) >= 4096
            and x.size(1) % 256 == 0 and _bf16_ok(x.size(1)))


def cached_bf16(x):
    """bf16 copy of a constant feature matrix, cached on the tensor object (rebuilt when it is written to)."""
    from .graph import _cached
    return _cached(x, "bf16", (), lambda: to_bf16(x))


def bf16_storage() -> bool:
    """bf16 mode also keeps the tables the gather kernels read in bf16 (fp32 accumulate)."""
    return _PRECISION == 1


def spmm(rowptr, col, w, x, n_rows, self_w=None, addend=None, bias=None, relu=False, drop: Dropout = None, tag=None,
         split=None, out_dtype=torch.float32, live=None, out_gate=None, out_gate_scale=1.0):
    """``out[i] = epi(sum_k w[k] x[col[k]] (+ self_w[i] x[i]) (+ addend[i]))`` (csrc/spmm.hip).
    ``x`` is an fp32 or a bf16 table; the result is fp32.  ``live``: device int32 count of the entries ``rowptr`` spans
    when ``col`` / ``w`` are longer than that (graph.KeptCSR) -- only the byte accounting of a tagged call reads it.
    ``out_gate`` (bf16 ``[n_rows, d]``, bf16 result only): the result leaves through the ReLU / dropout backward gate of
    the layer whose output ``out_gate`` is -- ``to_bf16(result, gate=out_gate, gate_scale=out_gate_scale)`` without the pass."""
    L.require_cuda(rowptr, col, w, x, self_w, addend, bias, out_gate)
    if out_gate is not None and (out_dtype != torch.bfloat16 or out_gate.dtype != torch.bfloat16 or not out_gate.is_contiguous()
                                 or out_gate.shape != (n_rows, x.size(1))):
        raise ValueError("spmm: out_gate is a contiguous bf16 [n_rows, d] table and goes with a bf16 result")
    if x.dtype == torch.bfloat16:
        x = x if x.is_contiguous() else x.contiguous()
    else:
        x = _f32c(x)
    is_bf16 = int(x.dtype == torch.bfloat16)
    d = x.size(1)
    out = torch.empty(n_rows, d, dtype=out_dtype, device=x.device)
    keep, p, seed, sid = None, 0.0, 0, 0
    if drop is not None and drop.p > 0.0:
        p, seed, sid = drop.p, drop.seed, drop.stream_id
        if drop.keep is not None:
            keep = drop.keep.to(torch.uint8).contiguous()
            if keep.shape != out.shape:
                raise ValueError("dropout keep mask shape mismatch")
    if addend is not None:       # fp32, or a bf16 table (a product only this epilogue reads: RGCNConv's root term)
        addend = (addend if addend.is_contiguous() else addend.contiguous()) if addend.dtype == torch.bfloat16 else _f32c(addend)
        if addend.dtype == torch.bfloat16 and (addend.stride(0) % 4 or addend.data_ptr() % 8):
            addend = addend.float()
    if split is not None:
        split.use()
    # rows of 1 KB (the two-view bf16 tables, the exact mode's fp32 tables) or 2 KB: the launch runs the row's 512-byte column
    # parts as two / four problems, each on four / two of the eight XCDs -- an L2 then caches pieces of a half / a quarter of
    # the table's bytes (csrc/spmm.hip, col_halves; measured on the bench graph: bf16 d = 512 352 -> 310 us, fp32 d = 512
    # 805 -> 588; 512-byte rows gain nothing from 256-byte parts)
    halves = 1
    if SPMM_COL_HALVES and split is not None and split.pack and n_rows >= 4096:
        row_bytes = d * x.element_size()
        halves = 4 if row_bytes >= 2048 else (2 if row_bytes >= 1024 else 1)
        if halves > 1 and (d % (8 * halves) or (d // halves) // (8 if is_bf16 else 4) not in (4, 8, 16, 32, 64)):
            halves = 1
    if split is not None and split.pack and (d // halves) // (8 if is_bf16 else 4) >= 64 and split.exact:
        split = split.plain()          # wide rows: one wave per row already (RowSplit.plain)
    timing = KERNEL_TIMERS is not None and tag is not None
    if timing:
        ev0, ev1 = _event(), _event()
        ev0.record()
    partials = split.partials(d, x.device) if split is not None else None
    L.call("bkg_spmm_csr_f32", L.ptr(rowptr), L.ptr(col), L.ptr(w), L.ptr(x), is_bf16, x.stride(0), L.ptr(out),
           int(out_dtype == torch.bfloat16), out.stride(0), n_rows, d, L.ptr(self_w), L.ptr(addend),
           int(addend is not None and addend.dtype == torch.bfloat16), addend.stride(0) if addend is not None else 0,
           L.ptr(bias), int(relu), L.ptr(keep), p, seed, sid, L.ptr(out_gate), out_gate.stride(0) if out_gate is not None else 0,
           float(out_gate_scale), L.sref(split.c if split is not None else None), L.ptr(partials), halves, L.stream())
    if timing:
        ev1.record()
        rec = KERNEL_TIMERS.setdefault(tag, {"events": [], "bytes": _spmm_bytes(col.numel(), n_rows, d, w is not None,
                                                                             2 if is_bf16 else 4, out.element_size()),
                                             # (entries by a device count: resolved after the timed loop, timer_bytes)
                                             "live": live, "bytes_of": (n_rows, d, w is not None, 2 if is_bf16 else 4,
                                                                        out.element_size()),
                                             "d": d, "col_halves": halves,
                                             # rocprofv3's Grid_Size of the two launches (threads)
                                             "grid": _spmm_grid(((split.n_items_max if split is not None else n_rows) + 3) // 4,
                                                                halves),
                                             "grid2": _spmm_second_grid(split, d // halves, is_bf16, halves)})
        rec["events"].append((ev0, ev1))
    return out


SPMM_COL_HALVES = __import__("os").environ.get("BKG_SPMM_COL_HALVES", "1") != "0"


def _spmm_grid(blocks, halves):
    """Threads of a launch of ``blocks`` workgroups (column parts: 2 | 4 problems, 4 | 2 blocks of each per group of eight)."""
    per = 8 // halves
    return ((blocks + per - 1) // per * 8 if halves > 1 else blocks) * 256


def _spmm_second_grid(split, d, is_bf16, halves=1):
    """Grid_Size (threads) of bkg_spmm_csr_f32's second launch -- long-row fix-up blocks + short-row blocks
    (csrc/spmm.hip launch_spmm / ShortShape) -- so that a PMC pass can be attributed to the tagged call."""
    if split is None:
        return None
    pieces = d // (8 if is_bf16 else 4)
    lpr, vpl = {96: (32, 3), 128: (64, 2), 192: (64, 3), 256: (64, 4)}.get(pieces, (min(pieces, 64), 1))
    c = pieces // 2
    ls = c if (pieces % 2 == 0 and 4 <= c <= 64 and 64 % c == 0) else lpr
    short_blocks = (split.n_short_max + 256 // ls - 1) // (256 // ls) if split.pack else 0
    return _spmm_grid(split.n_long_max + short_blocks, halves)


def _table(x):
    """The table a gather kernel reads: a bf16 copy in bf16 mode (when the shape allows 16-byte pieces)."""
    if bf16_storage() and x.dtype == torch.float32 and x.dim() == 2 and _bf16_ok(x.size(1)):
        return to_bf16(x)
    return x


def _bf16_ok(d):
    """Row widths the bf16-table SpMM kernels are instantiated for (d / 8 sixteen-byte pieces per row)."""
    return d % 8 == 0 and (d // 8) in (4, 8, 16, 32, 64, 96, 128, 192, 256)


def _plain_dropout_keep(relu, drop):
    """Dropout WITHOUT a ReLU in front (not on the reference's path: encoder.py:50-59,153-162 always pair them):
    the backward gate cannot be read off the output's sign -- a kept element may be exactly 0 -- so it needs the
    keep decisions themselves.  An explicit mask is carried to the backward; the in-kernel Philox stream is refused."""
    if relu or drop is None or drop.p <= 0:
        return None
    if drop.keep is None:
        raise NotImplementedError("dropout without ReLU needs an explicit keep mask (Dropout.keep)")
    return drop.keep


def _gate(dout, out, scale):
    """Backward of ``dropout(relu(.))`` from the saved OUTPUT: ``out > 0`` iff pre > 0 and kept."""
    if (dout.dtype == torch.float32 and out.dtype == torch.float32 and dout.is_contiguous() and out.is_contiguous()
            and dout.numel() % 4 == 0 and dout.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0):
        g = torch.empty_like(dout)
        L.call("bkg_gate_f32", L.ptr(dout), L.ptr(out), float(scale), dout.numel(), L.ptr(g), L.stream())
        return g
    g = torch.ops.aten.threshold_backward(dout, out, 0.0)        # dout where out > 0, else 0
    return g.mul_(scale) if scale != 1.0 else g


# ----------------------------------------------------------------------------- Linear
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        out = gemm(x, weight, trans_b=True, bias=bias, act=act)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, out if act != ACT_NONE else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, out = ctx.saved_tensors
        dout = _f32c(dout)
        if ctx.act == ACT_RELU:
            dout = _gate(dout, out, 1.0)
        elif ctx.act == ACT_ELU:  # elu'(pre) = 1 (pre > 0) else elu(pre) + 1
            dout = torch.where(out > 0, dout, dout * (out + 1.0))
        dx = None
        if ctx.needs_input_grad[0]:
            if x.dtype == torch.bfloat16 and _xw_ok(dout, dout.size(0), weight.size(1), weight.size(0)):
                dx = gemm(dout, weight, out_dtype=torch.bfloat16)        # the gradient of a bf16 activation is bf16
            else:
                dx = gemm(dout, weight).to(x.dtype)
        dw = None
        if ctx.needs_input_grad[1]:
            small = dout.size(0) < 4096 and x.dtype == torch.bfloat16
            dw = gemm(dout, x.float() if small else x, trans_a=True)
        db = colsum(dout) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


def linear(x, weight, bias=None, act=ACT_NONE):
    """``act(x @ weight^T + bias)`` with ``weight [out, in]`` (torch.nn.Linear layout)."""
    lead = x.shape[:-1]
    y = _Linear.apply(x.reshape(-1, x.size(-1)), weight, bias, act)
    return y.reshape(*lead, weight.size(0))


class _RowDot(torch.autograd.Function):
    """``s[i] = x[i, :] . w`` over a tall fp32 / bf16 table (csrc/rowdot.hip); backward: ``dx = g (x) w`` as a table of
    x's type, ``dw = sum_i g[i] x[i, :]`` (column sums with row weights, csrc/colsum.hip)."""

    @staticmethod
    def forward(ctx, x, w, link_in=None):
        out = torch.empty(x.size(0), dtype=torch.float32, device=x.device)
        L.call("bkg_rowdot_f32", L.ptr(x), int(x.dtype == torch.bfloat16), x.stride(0), x.size(0), x.size(1), L.ptr(w),
               L.ptr(out), L.stream())
        ctx.save_for_backward(x, w)
        ctx.link_in = link_in
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        n, d = x.shape
        g = _f32c(g)
        dx = dw = None
        li = ctx.link_in
        if (ctx.needs_input_grad[0] and GATE_IN_PRODUCT and li is not None and li.relu and x.dtype == torch.bfloat16
                and n > 4096 and d % 8 == 0 and d // 8 <= 256 and x.is_contiguous() and x.data_ptr() % 16 == 0
                and w.data_ptr() % 16 == 0):
            # x is the layer below's dropout(relu(.)) output: dx = g (x) w leaves through that layer's backward gate, with its
            # bias gradient (GateLink) -- neither the outer product nor its gated copy is a pass of its own
            dx = torch.empty(n, d, dtype=torch.bfloat16, device=x.device)
            ws = torch.empty(L.lib().bkg_colsum_workspace_floats(n, d), dtype=torch.float32, device=x.device)
            cs = torch.empty(d, dtype=torch.float32, device=x.device)
            L.call("bkg_outer_gate_colsum_bf16", L.ptr(g), L.ptr(w), L.ptr(x), float(li.scale), L.ptr(dx), n, d, L.ptr(ws),
                   L.ptr(cs), L.stream())
            li.applied, li.db = True, cs
        elif ctx.needs_input_grad[0]:
            dx = torch.empty(n, d, dtype=x.dtype, device=x.device)
            L.call("bkg_scaled_rows", L.ptr(g), L.ptr(w), n, d, L.ptr(dx), int(x.dtype == torch.bfloat16), dx.stride(0),
                   L.stream())
        if ctx.needs_input_grad[1]:
            ws = torch.empty(L.lib().bkg_colsum_workspace_floats(n, d), dtype=torch.float32, device=x.device)
            dw = torch.empty(d, dtype=torch.float32, device=x.device)
            L.call("bkg_colsum_weighted_f32", L.ptr(x), int(x.dtype == torch.bfloat16), x.stride(0), n, d, L.ptr(g), L.ptr(ws),
                   L.ptr(dw), L.stream())
        return dx, dw, None


class _ColumnsOfPair(torch.autograd.Function):
    """``s [n, 2] -> (s[:, 0], s[:, 1])``.  Autograd's own column selects hand back two zero-filled ``[n, 2]`` tables, two
    copies and their sum; the loss behind (``ggd_bce``) writes ``[d pos | d neg]`` as ONE ``[n, 2]`` table already -- its
    two columns arrive here and that table is the gradient (no launch), anything else is stacked (one)."""

    @staticmethod
    def forward(ctx, s):
        return s[:, 0], s[:, 1]

    @staticmethod
    def backward(ctx, g0, g1):
        b = g0._base
        if (b is not None and b is g1._base and b.dim() == 2 and b.size(1) == 2 and b.is_contiguous() and b.dtype == g0.dtype
                and g0.stride() == (2,) and g1.stride() == (2,) and g0.storage_offset() == b.storage_offset()
                and g1.storage_offset() == b.storage_offset() + 1 and g0.numel() == b.size(0)):
            return b
        return torch.stack((g0, g1), dim=1)


def columns_of_pair(s):
    """The two columns of an ``[n, 2]`` score table (GGD: positive | corrupted) as separate vectors."""
    return _ColumnsOfPair.apply(s)


class _ColSumOfWeight(torch.autograd.Function):
    """``weight.sum(0)`` of a small parameter through the library's column sum (the gradient: every row gets it)."""

    @staticmethod
    def forward(ctx, weight):
        ctx.rows = weight.size(0)
        return colsum(weight)

    @staticmethod
    def backward(ctx, g):
        return g.unsqueeze(0).expand(ctx.rows, g.numel())


def weight_colsum(weight):
    """``weight.sum(0)`` of a small 2-D parameter (or an expression of parameters) through the library's column sum,
    differentiable."""
    L.require_cuda(weight)
    return _ColSumOfWeight.apply(_f32c(weight))


def rowsum_linear_ok(x, weight):
    """Can ``linear(x, weight, bias).sum(-1)`` run as one dot product per row (:func:`linear_rowsum`)?"""
    epl = 8 if x.dtype == torch.bfloat16 else 4
    d = x.size(-1)
    return (x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16) and x.stride(1) == 1 and d % epl == 0
            and d // epl <= 256 and (d // epl) in _KGE_PIECES and x.stride(0) % epl == 0 and x.data_ptr() % 16 == 0
            and weight.dtype == torch.float32)


def row_dot_ok(d, dtype=torch.float32):
    """Row widths ``row_dot`` takes for a contiguous table of ``dtype``."""
    epl = 8 if dtype == torch.bfloat16 else 4
    return d % epl == 0 and d // epl <= 256 and (d // epl) in _KGE_PIECES


def row_dot(x, w, link_in=None):
    """``s[i] = x[i, :] . w`` over a tall contiguous fp32 / bf16 table (csrc/rowdot.hip), differentiable in both.
    ``link_in`` (``GateLink``): ``x`` is the output of the conv layer that holds the other end of the link."""
    L.require_cuda(x, w)
    if x.dim() != 2 or not row_dot_ok(x.size(1), x.dtype) or x.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("row_dot: a contiguous fp32 / bf16 [rows, d] table of a supported width (row_dot_ok)")
    return _RowDot.apply(x if x.is_contiguous() else x.contiguous(), _f32c(w), link_in)


def linear_rowsum(x, weight, bias=None):
    """``linear(x, weight, bias).sum(1)`` -- the head of GGD, reference biomedkg/model/gcl.py:86-90 -- without the
    ``[rows, out]`` product: ``sum_j (W x + b)_j = x . (sum_j W[j, :]) + sum_j b_j``.  Same value up to the order of
    the fp32 additions (so the exact-fp32 parity path keeps the product); the gradients reach ``weight`` and ``bias``
    through autograd of the two small sums."""
    L.require_cuda(x, weight, bias)
    s = _RowDot.apply(x if x.is_contiguous() else x.contiguous(), _ColSumOfWeight.apply(weight))
    return s if bias is None else s + bias.sum()


# ----------------------------------------------------------------------------- GCNConv
def _hidden_bf16_ok(n_rows, d):
    """bf16 mode keeps HIDDEN layer outputs (everything between two conv layers) as bf16 tables: each is only ever
    an MFMA operand, a gather table or the sign gate of the backward, all of which round it to bf16 anyway."""
    return bf16_storage() and n_rows >= 4096 and d % 256 == 0 and _bf16_ok(d)


def _as_f32(t):
    return t if t.dtype == torch.float32 else t.float()


def _dims(graph):
    """(output rows, gather-table rows, exchange) of a message-passing structure: a whole graph has
    ``n_out == n_table`` and no exchange; a node-partitioned one (biomedkg_amd/partition.py) owns ``n_out`` target rows,
    gathers from the all-gathered ``n_table``-row table and carries the collective pair in ``comm``."""
    n = graph.num_nodes
    comm = getattr(graph, "comm", None)
    if comm is not None and not comm.live:
        comm = None                        # one rank: nothing to exchange, no fp32 partial tables needed
    return getattr(graph, "n_out", n), getattr(graph, "n_table", n), comm


def _exchanged_product(comm, n_rows, d_out, dtype, device, make_rows, min_rows=0):
    """A node-partitioned layer's gather table: ``make_rows(lo, hi, out)`` writes rows ``[lo, hi)`` of the rank's product
    into ``out``; every finished row chunk leaves for its readers while the next one is computed (halo patterns; the
    plain all-gather moves the whole shard after the last chunk).  -> the ``[world * n_max, d_out]`` table."""
    buf = comm.shard_buffer(d_out, dtype, device)
    chunks = comm.chunks("in")
    # a chunk the producer's kernel does not take on its own (fp32 rows below the streaming product's floor: a rank that
    # owns hub nodes holds few rows, and a quarter of them fewer still): the whole shard as ONE product, then the chunked
    # exchange all the same (every rank must issue the same collectives)
    whole = any(0 < hi - lo < min_rows for lo, hi in chunks)
    if whole and n_rows > 0:
        make_rows(0, n_rows, buf[:n_rows])
    pending = []
    for c, (lo, hi) in enumerate(chunks):
        if hi > lo and not whole:
            make_rows(lo, hi, buf[lo:hi])
        pending.append(comm.exchange_begin(buf, "in", chunk=c if len(chunks) > 1 else None))
    table = None
    for p in pending:
        table = comm.exchange_end(p)
    return table


def _view_csr(gw, side):
    """The ``dropout_edge`` view's own CSR (graph.KeptCSR) when the mask was compacted, else None (the graph's CSR with
    the view's weights: partitioned graphs, mini-batches, no mask)."""
    return getattr(gw, side, None)


def _view_tag(tag, csr):
    """Launches over a view's kept entries are timed apart from the full graph's (other bytes per launch, other grid)."""
    return tag + "_kept" if isinstance(csr, KeptCSR) else tag


class _GcnConv(torch.autograd.Function):
    """``dropout(relu(A_hat (x W^T) + b))`` -- PyG GCNConv + the encoder's ReLU/dropout
    (reference biomedkg/model/encoder.py:153-162) fused into the aggregation epilogue.  ``x`` is fp32 or (a hidden
    activation of the bf16 mode) bf16; ``hidden`` asks for a bf16 result.  Node-partitioned graphs: the product runs on
    the rank's rows, the table is all-gathered (transform BEFORE the exchange: the wire carries ``out`` columns), the
    backward's table gradient is reduce-scattered to its owners."""

    @staticmethod
    def forward(ctx, x, weight, bias, graph: Graph, gw: GcnWeights, relu, drop, hidden):
        n_rows, n_table, comm = _dims(graph)
        d_out, d_in = weight.size(0), weight.size(1)
        if x.dtype == torch.bfloat16 and not _xw_ok(x, n_rows, d_out, d_in):
            x = x.float()                      # shapes the streaming product does not take: widen once
        if bf16_storage() and _bf16_ok(d_out) and _xw_ok(x, n_rows, d_out, d_in):
            if comm is not None:          # row chunks of the product leave for their readers while the next one runs
                xw = _exchanged_product(comm, n_rows, d_out, torch.bfloat16, x.device,
                                        lambda lo, hi, out: gemm(x[lo:hi], weight, trans_b=True, out_dtype=torch.bfloat16, out=out),
                                        min_rows=0 if x.dtype == torch.bfloat16 else _TALL_ROWS)
            else:
                xw = gemm(x, weight, trans_b=True, out_dtype=torch.bfloat16)      # only ever read as the gather table
        else:
            xw = _table(gemm(x, weight, trans_b=True))
            if comm is not None:
                xw = comm.exchange(xw, "in")
        c = _view_csr(gw, "dst") or graph.by_dst
        out16 = hidden and _hidden_bf16_ok(n_rows, d_out)
        out = spmm(c.rowptr, c.col, gw.w_dst, xw, n_rows, self_w=gw.self_w, bias=bias, relu=relu, drop=drop,
                   tag=_view_tag("gcn_propagate", c), split=c.split, out_dtype=torch.bfloat16 if out16 else torch.float32,
                   live=getattr(c, "count", None))
        ctx.graph, ctx.gw, ctx.relu = graph, gw, relu
        ctx.scale = 1.0 / (1.0 - drop.p) if (drop is not None and drop.p > 0) else 1.0
        ctx.gated = relu or ctx.scale != 1.0
        ctx.keep = _plain_dropout_keep(relu, drop)
        ctx.save_for_backward(x, weight, out if ctx.gated else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, out = ctx.saved_tensors
        g, gw = ctx.graph, ctx.gw
        n_rows, n_table, comm = _dims(g)
        dout = dout if dout.is_contiguous() else dout.contiguous()
        db = None
        if bf16_storage() and (ctx.relu or not ctx.gated) and _bf16_ok(dout.size(1)):
            # gate + bf16 gather table (+ the bias gradient) in one pass over dOut (no fp32 gated copy is ever written)
            if ctx.needs_input_grad[2]:
                dout, db = to_bf16_colsum(dout, gate=out if ctx.gated else None, gate_scale=ctx.scale)
            else:
                dout = to_bf16(dout, gate=out if ctx.gated else None, gate_scale=ctx.scale)
        elif ctx.gated:
            dout, out = _as_f32(dout), _as_f32(out)
            dout = _gate(dout, out, ctx.scale) if ctx.relu else dout * ctx.keep * ctx.scale
        else:
            dout = _as_f32(dout)
        c = _view_csr(gw, "src") or g.by_src
        # dXW feeds only the two products below: keep it as a bf16 table when both can take one
        d_out, d_in = weight.size(0), weight.size(1)
        x16 = x.dtype == torch.bfloat16
        to16 = (bf16_storage() and _bf16_ok(d_out) and n_rows >= 4096
                and ((d_out + 63) // 64) * ((d_in + 63) // 64) <= 256
                and (not ctx.needs_input_grad[0] or _bf16_operand_ok(n_rows, d_in, d_out)))
        if comm is not None:
            # node-partitioned: the (gated, bf16) output gradient goes to the ranks whose out-edges point at it, and this
            # rank sums the table gradient of ITS OWN rows over its out-edge CSR -- the single-GPU sum, no reduce-scatter
            dout_t, rows = comm.exchange(dout, "out"), n_rows
        else:
            dout_t, rows = dout, n_table
        dxw = spmm(c.rowptr, c.col, gw.w_src, dout_t, rows, self_w=gw.self_w, split=c.split,
                   out_dtype=torch.bfloat16 if to16 else torch.float32)
        dx = None
        if ctx.needs_input_grad[0]:
            if x16 and _xw_ok(dxw, n_rows, d_in, d_out):
                dx = gemm(dxw, weight, out_dtype=torch.bfloat16)     # the gradient of a bf16 activation is bf16
            else:
                dx = gemm(dxw, weight).to(x.dtype)
        if ctx.needs_input_grad[1]:
            if (x16 or dxw.dtype == torch.bfloat16) and n_rows < 4096:
                dw = gemm(_as_f32(dxw), _as_f32(x), trans_a=True)
            else:
                dw = gemm(dxw, x, trans_a=True)
        else:
            dw = None
        if db is None and ctx.needs_input_grad[2]:
            db = colsum(dout)
        return dx, dw, db, None, None, None, None, None


def gcn_conv(x, weight, bias, graph, gw, relu=False, drop=None, hidden=False):
    L.require_cuda(x, weight, bias)
    if x.dtype != torch.bfloat16:
        x = _f32c(x)
    return _GcnConv.apply(x, weight, bias, graph, gw, relu, drop, hidden)


class _XwTable(torch.autograd.Function):
    """``bf16(x W^T)`` as a gather table shared by several views (GGD / DGI corrupt the input by a row permutation,
    and ``x[perm] W^T = (x W^T)[perm]``: the first layer's product is computed once)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        if not bf16_storage():          # exact mode: an fp32 table (each row's product does not depend on where the row sits)
            return gemm(x, weight, trans_b=True)
        return gemm(x, weight, trans_b=True, out_dtype=torch.bfloat16)

    @staticmethod
    def backward(ctx, dxw):
        x, weight = ctx.saved_tensors
        dxw = dxw if dxw.is_contiguous() else dxw.contiguous()
        dx = gemm(dxw, weight, out_dtype=x.dtype) if ctx.needs_input_grad[0] else None
        dw = gemm(dxw, x, trans_a=True) if ctx.needs_input_grad[1] else None
        return dx, dw


def xw_table(x, weight):
    L.require_cuda(x, weight)
    return _XwTable.apply(x if x.dtype == torch.bfloat16 else _f32c(x), weight)


class _PairRows(torch.autograd.Function):
    """``[x | x[perm]]`` row by row for a bf16 table (csrc/pair.hip); the gradient gathers through the inverse perm."""

    @staticmethod
    def forward(ctx, x, perm):
        n, h = x.shape
        out = torch.empty(n, 2 * h, dtype=torch.bfloat16, device=x.device)
        L.call("bkg_pair_rows_bf16", L.ptr(x), L.ptr(perm), L.ptr(out), n, h, L.stream())
        ctx.perm = perm
        return out

    @staticmethod
    def backward(ctx, g):
        perm = ctx.perm
        n, h = g.size(0), g.size(1) // 2
        g = g if (g.dtype == torch.bfloat16 and g.is_contiguous()) else g.to(torch.bfloat16).contiguous()
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(n, device=perm.device)
        out = torch.empty(n, h, dtype=torch.bfloat16, device=g.device)
        L.call("bkg_unpair_rows_bf16", L.ptr(g), L.ptr(inv), L.ptr(out), n, h, L.stream())
        return out, None


def pair_rows(x, perm):
    """``[N, 2h]`` bf16: row i = ``[x[i] | x[perm[i]]]`` (``perm`` int64, a permutation of the rows)."""
    L.require_cuda(x, perm)
    if x.dtype != torch.bfloat16 or x.size(1) % 8 or not x.is_contiguous():
        return torch.stack([x, x[perm]], dim=1).view(x.size(0), 2 * x.size(1))
    return _PairRows.apply(x, perm.contiguous())


def views_ok(n_rows, d_in, d_out, views=2):
    """Shapes for which V views of one graph run through a conv layer as ONE launch per kernel (bf16 mode)."""
    return (bf16_storage() and n_rows >= 4096 and d_in % 256 == 0 and d_out % 256 == 0 and _bf16_ok(views * d_out)
            and _bf16_operand_ok(views * n_rows, d_out, d_in) and _bf16_operand_ok(views * n_rows, d_in, d_out))


FP32_VIEWS = __import__("os").environ.get("BKG_FP32_VIEWS", "1") != "0"


def views_ok_f32(n_rows, d_in, d_out, views=2):
    """Exact-fp32 mode: V views of one graph through a conv layer side by side (``_GcnConvViewsF32``) -- the row widths the
    fp32 aggregation kernels take (``views * d_out / 4`` sixteen-byte pieces, in column parts above 1 KB rows)."""
    d = views * d_out
    return (FP32_VIEWS and _PRECISION == 0 and n_rows >= 4096 and d_in % 4 == 0 and d % 16 == 0
            and (d // 4 in (4, 8, 16, 32, 64, 96, 128, 192, 256) or (d % 64 == 0 and d // 16 in (4, 8, 16, 32, 64))))


class _GcnConvViewsF32(torch.autograd.Function):
    """``_GcnConvViews`` in the exact-fp32 mode (whole graphs and mini-batches on one GPU): the V views' rows of a node side
    by side in fp32 tables -- one product over ``[V N, in]`` rows, one aggregation launch over ``V * out``-wide rows (its
    index / weight stream read once), one gate and one pair of gradient products per layer instead of V.  Every row's
    arithmetic is what the one-view layer does to it (the products are row-independent, the aggregation sums a row's
    entries column by column), so the corrupted view's first product ``x[perm] W^T = (x W^T)[perm]`` is shared exactly."""

    @staticmethod
    def forward(ctx, x2, weight, bias, graph: Graph, gw: GcnWeights, relu, drop, views, is_table):
        n, n_table, comm = _dims(graph)
        if comm is not None:
            raise ValueError("the fp32 side-by-side layer runs whole graphs (a partitioned exact-mode step keeps one view per pass)")
        x2 = _f32c(x2)
        if is_table:
            table = x2
        else:
            d_in, d_out = weight.size(1), weight.size(0)
            table = gemm(x2.view(views * n, d_in), weight, trans_b=True).view(n, views * d_out)
        c = _view_csr(gw, "dst") or graph.by_dst
        out = spmm(c.rowptr, c.col, gw.w_dst, table, n, self_w=gw.self_w, bias=bias.repeat(views), relu=relu, drop=drop,
                   split=c.split)
        ctx.graph, ctx.gw, ctx.relu, ctx.views, ctx.is_table = graph, gw, relu, views, is_table
        ctx.scale = 1.0 / (1.0 - drop.p) if (drop is not None and drop.p > 0) else 1.0
        ctx.gated = relu or ctx.scale != 1.0
        ctx.keep = _plain_dropout_keep(relu, drop)
        ctx.save_for_backward(None if is_table else x2, weight, out if ctx.gated else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, weight, out = ctx.saved_tensors
        g, gw, v = ctx.graph, ctx.gw, ctx.views
        n, n_table, _ = _dims(g)
        dout = _f32c(dout)
        if ctx.gated:
            dout = _gate(dout, out, ctx.scale) if ctx.relu else dout * ctx.keep * ctx.scale
        d_out = dout.size(1) // v
        db = colsum(dout.view(v * n, d_out)) if ctx.needs_input_grad[2] else None       # the views share the bias
        c = _view_csr(gw, "src") or g.by_src
        dxw = spmm(c.rowptr, c.col, gw.w_src, dout, n_table, self_w=gw.self_w, split=c.split)
        if ctx.is_table:
            return dxw, None, db, None, None, None, None, None, None
        d_in = weight.size(1)
        rows = dxw.view(v * n, d_out)
        dx = gemm(rows, weight).view(n, v * d_in) if ctx.needs_input_grad[0] else None
        dw = gemm(rows, x2.view(v * n, d_in), trans_a=True) if ctx.needs_input_grad[1] else None
        return dx, dw, db, None, None, None, None, None, None


class _GcnConvViews(torch.autograd.Function):
    """V views of the SAME graph through one GCNConv (GGD: positive + corrupted input).  ``x2`` is ``[N, V * in]`` --
    row i holds the V views' rows of node i side by side -- so the products see ``[V * N, in]`` (one launch, the
    weight slab loaded once, one weight-gradient product over K = V * N) and the aggregation gathers ``V * out``-wide
    rows with one index / weight stream.  ``is_table``: ``x2`` already is the gather table (shared first-layer
    product, see ``_XwTable``; on a node-partitioned graph: the ALL-GATHERED table) and ``weight`` is unused.
    bf16 mode only (``views_ok``).  Node-partitioned graphs exchange the table as ``_GcnConv`` does."""

    @staticmethod
    def forward(ctx, x2, weight, bias, graph: Graph, gw: GcnWeights, relu, drop, hidden, views, is_table, link_in=None,
                link_out=None):
        n, n_table, comm = _dims(graph)
        ctx.link_in, ctx.link_out = link_in, link_out
        if is_table:
            table = x2 if x2.is_contiguous() else x2.contiguous()
        else:
            d_in, d_out = weight.size(1), weight.size(0)
            if comm is not None:
                xr = x2.view(views * n, d_in)
                table = _exchanged_product(comm, n, views * d_out, torch.bfloat16, x2.device,
                                           lambda lo, hi, out: gemm(xr[views * lo:views * hi], weight, trans_b=True,
                                                                    out_dtype=torch.bfloat16,
                                                                    out=out.view(views * (hi - lo), d_out)),
                                           min_rows=0 if x2.dtype == torch.bfloat16 else (_TALL_ROWS + views - 1) // views)
            else:
                table = gemm(x2.view(views * n, d_in), weight, trans_b=True, out_dtype=torch.bfloat16).view(n, views * d_out)
        c = _view_csr(gw, "dst") or graph.by_dst
        out = spmm(c.rowptr, c.col, gw.w_dst, table, n, self_w=gw.self_w, bias=bias.repeat(views), relu=relu, drop=drop,
                   tag=_view_tag("gcn_propagate_views", c), split=c.split, out_dtype=torch.bfloat16 if hidden else torch.float32,
                   live=getattr(c, "count", None))
        ctx.graph, ctx.gw, ctx.relu, ctx.views, ctx.is_table = graph, gw, relu, views, is_table
        ctx.scale = 1.0 / (1.0 - drop.p) if (drop is not None and drop.p > 0) else 1.0
        ctx.gated = relu or ctx.scale != 1.0
        ctx.keep = _plain_dropout_keep(relu, drop)
        ctx.table_dtype = x2.dtype
        if link_out is not None:
            link_out.relu, link_out.scale = bool(relu), ctx.scale
        ctx.save_for_backward(None if is_table else x2, weight, out if ctx.gated else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, weight, out = ctx.saved_tensors
        g, gw, v = ctx.graph, ctx.gw, ctx.views
        n, n_table, comm = _dims(g)
        dout = dout if dout.is_contiguous() else dout.contiguous()
        cs = None
        lo = ctx.link_out
        if lo is not None and lo.applied:      # the layer above gated its product's result and summed its columns
            cs, lo.db, lo.applied = lo.db, None, False     # consumed: valid for THIS backward only
            if dout.dtype != torch.bfloat16:   # (never gated twice, whatever autograd did to the tensor on its way here)
                dout = to_bf16(dout)
        elif ctx.relu or not ctx.gated:
            if ctx.needs_input_grad[2]:
                dout, cs = to_bf16_colsum(dout, gate=out if ctx.gated else None, gate_scale=ctx.scale)
            else:
                dout = to_bf16(dout, gate=out if ctx.gated else None, gate_scale=ctx.scale)
        else:
            dout = to_bf16(_as_f32(dout) * ctx.keep * ctx.scale)
        c = _view_csr(gw, "src") or g.by_src
        d_out = dout.size(1) // v
        db = None
        if ctx.needs_input_grad[2]:        # the views share the bias: their column sums added
            if cs is not None and cs.numel() == d_out:
                db = cs                    # (summed over the rows of all views already)
            else:
                db = cs.view(v, d_out).sum(0) if cs is not None else colsum(dout.view(v * n, d_out))
        if comm is not None:               # gather dOut, sum the table gradient of the rank's own rows (see _GcnConv)
            dxw = spmm(c.rowptr, c.col, gw.w_src, comm.exchange(dout, "out"), n, self_w=gw.self_w, split=c.split,
                       out_dtype=torch.bfloat16)
            if ctx.is_table:               # a gathered input table: only the rank's own slots carry a gradient
                full = torch.zeros(n_table, dxw.size(1), dtype=ctx.table_dtype, device=dxw.device)
                comm.own_rows(full)[:n] = dxw
                return full, None, db, None, None, None, None, None, None, None, None, None
        else:
            dxw = spmm(c.rowptr, c.col, gw.w_src, dout, n_table, self_w=gw.self_w, split=c.split, out_dtype=torch.bfloat16,
                       tag=_view_tag("gcn_propagate_views_bwd", c), live=getattr(c, "count", None))
            if ctx.is_table:
                return dxw.to(ctx.table_dtype), None, db, None, None, None, None, None, None, None, None, None
        d_in = weight.size(1)
        dxw_rows = dxw.view(v * n, d_out)
        dx, li = None, ctx.link_in
        if ctx.needs_input_grad[0]:
            x_rows = x2.view(v * n, d_in)
            if (li is not None and li.relu and comm is None and d_in == d_out
                    and gemm_gated_ok(dxw_rows, weight, x_rows)):
                # x2 is the layer below's dropout(relu(.)) output: its sign is that layer's backward gate
                dx, li.db = gemm_gated(dxw_rows, weight, x_rows, li.scale)
                li.applied = True
                dx = dx.view(n, v * d_in)
            else:
                dx = gemm(dxw_rows, weight, out_dtype=x2.dtype).view(n, v * d_in)
        dw = gemm(dxw_rows, x2.view(v * n, d_in), trans_a=True) if ctx.needs_input_grad[1] else None
        return dx, dw, db, None, None, None, None, None, None, None, None, None


def gcn_conv_views(x2, weight, bias, graph, gw, views, relu=False, drop=None, hidden=False, is_table=False, link_in=None,
                   link_out=None):
    """``link_in`` / ``link_out`` (``GateLink``, bf16 mode): the hand-over with the layer below / above in a stack."""
    L.require_cuda(x2, weight, bias)
    if not bf16_storage():          # exact mode: fp32 tables throughout (``hidden`` asks for bf16 results: not here)
        return _GcnConvViewsF32.apply(x2, weight, bias, graph, gw, relu, drop, views, is_table)
    return _GcnConvViews.apply(x2, weight, bias, graph, gw, relu, drop, hidden, views, is_table, link_in, link_out)


class _GcnPropagateNarrow(torch.autograd.Function):
    """``A_hat u`` for a FEW columns (``u [N, c]``, c <= 16, fp32): the aggregation of a conv layer whose product has been
    folded into what follows it (``gcn_propagate_narrow``).  The columns ride in a 64-byte-row fp32 table (8 MB at the
    bench size: it lives in the caches); the backward is the same aggregation over the out-edge CSR."""

    @staticmethod
    def forward(ctx, u, graph: Graph, gw: GcnWeights):
        n, c = u.shape
        t = torch.zeros(n, 16, dtype=torch.float32, device=u.device)
        t[:, :c] = u
        cf = _view_csr(gw, "dst") or graph.by_dst
        out = spmm(cf.rowptr, cf.col, gw.w_dst, t, n, self_w=gw.self_w, split=cf.split)
        ctx.graph, ctx.gw, ctx.c = graph, gw, c
        return out[:, :c].contiguous()

    @staticmethod
    def backward(ctx, g):
        n, c = g.shape
        t = torch.zeros(n, 16, dtype=torch.float32, device=g.device)
        t[:, :c] = g
        cb = _view_csr(ctx.gw, "src") or ctx.graph.by_src
        d = spmm(cb.rowptr, cb.col, ctx.gw.w_src, t, n, self_w=ctx.gw.self_w, split=cb.split)
        return d[:, :c].contiguous(), None, None


def gcn_propagate_narrow(u, graph, gw):
    """``A_hat u`` (GCNConv's normalised aggregation) of ``u [N, c]``, ``c <= 16`` fp32 columns, differentiable
    in ``u``.  Whole graphs on one GPU."""
    L.require_cuda(u)
    if u.dim() != 2 or u.size(1) > 16 or u.dtype != torch.float32 or _dims(graph)[2] is not None:
        raise ValueError("gcn_propagate_narrow: an fp32 [N, c <= 16] table over a whole graph")
    return _GcnPropagateNarrow.apply(u.contiguous(), graph, gw)


# GGD's last conv layer folded into its head (model/gcl.py GGD._forward_views): 0 keeps the layer's [N, out] aggregation
GGD_FOLD_LAST = __import__("os").environ.get("BKG_GGD_FOLD_LAST", "1") != "0"


# ----------------------------------------------------------------------------- RGCNConv
_BRANCH_ROWS = 8192     # layers over fewer rows than this are launch-bound: independent chains go to a side stream
#                         (whole-graph layers, bandwidth-bound kernels both: 9.335 against 9.337 ms per step, round 5)
_SIDE_STREAMS = {}


def _side_stream(device):
    key = str(device)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class _RgcnConv(torch.autograd.Function):
    """``dropout(relu(sum_r mean_{N_r(i)}(x_j) W_r + x_i W_root + b))`` -- PyG RGCNConv
    + the encoder's ReLU/dropout (reference biomedkg/model/encoder.py:50-59).  ``x`` is fp32 or (a hidden activation
    of the bf16 mode) bf16; ``hidden`` asks for a bf16 result.  Node-partitioned graphs: the layer INPUT is the
    all-gathered table (started first, the root product runs under it); the backward's table gradient is
    reduce-scattered while the root / weight-gradient products run."""

    @staticmethod
    def forward(ctx, x, weight, root, bias, rg: RelGraph, relu, drop, hidden, training_pass=True, link_in=None, link_out=None):
        n, n_table, comm = _dims(rg)
        d_in, d_out = weight.size(1), weight.size(2)
        ctx.link_in, ctx.link_out = link_in, link_out
        x_given = x                        # (the caller's tensor object: what a remembered table is valid for)
        if x.dtype == torch.bfloat16 and not _xw_ok(x, n, d_out, d_in):
            x = x.float()
        remember = False
        if comm is None:
            table = _table(x)
            pending = None
        elif not ctx.needs_input_grad[0] and comm.constant_table(rg, x_given, ("in", _PRECISION)) is not None:
            table, pending = comm.constant_table(rg, x_given, ("in", _PRECISION)), None      # the node features: the same rows every step, gathered once
        else:
            remember = comm.is_constant_input(ctx.needs_input_grad, training_pass)
            pending = comm.exchange_begin(_table(x), "in")                           # crosses xGMI ...
        # ... under the [N, out] root term: a bf16 table where the layer's own output is one (a hidden layer of bf16 mode:
        # every value of the sum is rounded to bf16 right after it), fp32 for the encoder's output layer
        xr16 = (hidden and _hidden_bf16_ok(n, d_out) and x.dtype == torch.bfloat16 and _xw_ok(x, n, d_out, d_in))
        xr = gemm(x, root, out_dtype=torch.bfloat16) if xr16 else gemm(x, root)
        if pending is not None:
            table = comm.exchange_end(pending)
            if remember:
                comm.remember_table(rg, x_given, ("in", _PRECISION), table)
        h16 = bf16_storage() and _bf16_ok(d_in) and _bf16_operand_ok(rg.num_segments, d_out, d_in)
        # The segment means of an input that needs no gradient and is the SAME tensor at the same version as last time --
        # the constant node features under the first layer of a whole-graph run (reference kge_module.py:77-84: random /
        # LM features behind a frozen fuser) -- are those of last time: kept on the graph, like the features' bf16 copy.
        h, keep_h = None, (RGCN_KEEP_SEGMENT_MEANS and comm is None and not ctx.needs_input_grad[0]
                           and not torch.cuda.is_current_stream_capturing())
        if keep_h:
            held = getattr(rg, "_bkg_seg_means", None)
            if (held is not None and held[0]() is x_given and held[1] == x_given._version and held[2] == (h16, _PRECISION)
                    and held[3].size(0) == rg.num_segments):
                h = held[3]
        if h is None:
            h = spmm(rg.seg_ptr, rg.src_sorted, rg.inv_cnt, table, rg.num_segments, tag="rgcn_segment_mean",
                     split=rg.seg_split, out_dtype=torch.bfloat16 if h16 else torch.float32)   # [S, in]  segment means
            if keep_h and training_pass:       # (an evaluation pass feeds every layer a gradient-free input: nothing to keep)
                rg._bkg_seg_means = (weakref.ref(x_given), x_given._version, (h16, _PRECISION), h)
        y16 = bf16_storage() and _bf16_ok(d_out) and _xw_ok(h, h.size(0), d_out, d_in)
        y = gemm_grouped_rows(h, weight, rg.rel_ptr,                                  # [S, out] per-relation W_r; only
                              out_dtype=torch.bfloat16 if y16 else torch.float32)     # read as a gather table
        c = rg.dst_csr
        out16 = hidden and _hidden_bf16_ok(n, d_out)
        out = spmm(c.rowptr, c.col, None, y, n, addend=xr, bias=bias, relu=relu, drop=drop, split=c.split,
                   out_dtype=torch.bfloat16 if out16 else torch.float32)
        ctx.rg, ctx.relu = rg, relu
        ctx.scale = 1.0 / (1.0 - drop.p) if (drop is not None and drop.p > 0) else 1.0
        ctx.gated = relu or ctx.scale != 1.0
        ctx.keep = _plain_dropout_keep(relu, drop)
        if link_out is not None:
            link_out.relu, link_out.scale = bool(relu), ctx.scale
        ctx.save_for_backward(x, weight, root, h, out if ctx.gated else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, root, h, out = ctx.saved_tensors
        rg = ctx.rg
        n, n_table, comm = _dims(rg)
        d_in, d_out = weight.size(1), weight.size(2)
        dout = dout if dout.is_contiguous() else dout.contiguous()
        db0 = None
        lo = ctx.link_out
        if lo is not None and lo.applied:
            # the layer above wrote dOut through this layer's gate (the epilogue of its last aggregation): a bf16 table
            lo.applied = False                 # consumed: valid for THIS backward only
            if not (bf16_storage() and _bf16_ok(d_out) and n >= 4096 and _bf16_operand_ok(n, d_in, d_out)):
                dout = _as_f32(dout)
        elif (bf16_storage() and (ctx.relu or not ctx.gated) and _bf16_ok(d_out) and n >= 4096
                and _bf16_operand_ok(n, d_in, d_out)):
            # gated dOut as one bf16 table: gather table of dY, operand of the root products, source of db
            if ctx.needs_input_grad[3]:
                dout, db0 = to_bf16_colsum(dout, gate=out if ctx.gated else None, gate_scale=ctx.scale)
            else:
                dout = to_bf16(dout, gate=out if ctx.gated else None, gate_scale=ctx.scale)
        else:
            dout = _as_f32(dout)
            if ctx.gated:
                out = _as_f32(out)
                dout = _gate(dout, out, ctx.scale) if ctx.relu else dout * ctx.keep * ctx.scale
        # dY[s] = dOut[dst_s] is never stored when both of its consumers can read dOut's rows through seg_dst
        gathered = (dout.dtype == torch.bfloat16 and gathered_k_ok(h, dout)
                    and _xw_ok(dout, rg.num_segments, d_in, d_out))
        dy = None
        if not gathered:
            dy16 = (bf16_storage() and _bf16_ok(d_out) and
                    (not ctx.needs_input_grad[0] or _bf16_operand_ok(rg.num_segments, d_in, d_out)))
            dy = spmm(rg.seg_iota_ptr, rg.seg_dst, None, dout, rg.num_segments,       # dY[s] = dOut[dst_s]; read by
                      out_dtype=torch.bfloat16 if dy16 else torch.float32)            # the two products below only
        def weight_grads():
            dweight = None
            if ctx.needs_input_grad[1]:
                dweight = (gemm_grouped_k(h, dout, rg.rel_ptr, rg.num_relations, b_rows=rg.seg_dst) if gathered
                           else gemm_grouped_k(h, dy, rg.rel_ptr, rg.num_relations))
            droot = None
            if ctx.needs_input_grad[2]:
                if n < 4096 and (x.dtype == torch.bfloat16 or dout.dtype == torch.bfloat16):
                    droot = gemm(_as_f32(x), _as_f32(dout), trans_a=True)
                else:
                    droot = gemm(x, dout, trans_a=True)
            db = (db0 if db0 is not None else colsum(dout)) if ctx.needs_input_grad[3] else None
            return dweight, droot, db
        # mini-batches: the three parameter gradients are a chain of small launches independent of the input-gradient
        # chain -- run them as a parallel branch (side stream; a parallel branch of a captured graph)
        branch = comm is None and n < _BRANCH_ROWS and ctx.needs_input_grad[0]
        if branch:
            main, side = torch.cuda.current_stream(), _side_stream(dout.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                dweight, droot, db = weight_grads()
        if comm is not None:
            # node-partitioned: dOut (bf16) goes to the ranks whose out-edges point at it while the weight / root / bias
            # gradients -- all local -- are computed; then the table gradient of the rank's OWN rows is regrouped from the
            # gathered dOut: (relation, source) segment sums with the forward's edge weights, products with W_r^T,
            # per-source sum + the root part.  The forward's three kernels over rg.rev; nothing is reduce-scattered.
            pending = comm.exchange_begin(dout, "out") if ctx.needs_input_grad[0] else None
            dweight, droot, db = weight_grads()
            dx = None
            if ctx.needs_input_grad[0]:
                x16 = x.dtype == torch.bfloat16 and _bf16_ok(d_in)
                b16 = dout.dtype == torch.bfloat16
                dx_root = gemm(dout, root, trans_b=True,
                               out_dtype=torch.bfloat16 if (x16 and b16 and _xw_ok(dout, n, d_in, d_out)) else torch.float32)
                dtable = comm.exchange_end(pending)
                rev = rg.rev
                t16 = b16 and bf16_storage() and _bf16_ok(d_out) and _bf16_operand_ok(rev.num_segments, d_in, d_out)
                t = spmm(rev.seg_ptr, rev.src_sorted, rev.inv_cnt, dtable, rev.num_segments, split=rev.seg_split,
                         out_dtype=torch.bfloat16 if t16 else torch.float32)
                s16 = bf16_storage() and _bf16_ok(d_in) and _xw_ok(t, t.size(0), d_in, d_out)
                dxs = gemm_grouped_rows(t, weight, rev.rel_ptr, trans_b=True,
                                        out_dtype=torch.bfloat16 if s16 else torch.float32)
                c = rev.dst_csr
                dx = spmm(c.rowptr, c.col, None, dxs, n, addend=dx_root, split=c.split,
                          out_dtype=torch.bfloat16 if x16 else torch.float32)
                if dx.dtype != x.dtype:
                    dx = dx.to(x.dtype)
            return dx, dweight, droot, db, None, None, None, None, None, None, None
        # the input gradient: on a partitioned graph its table part goes out while the weight gradients run
        dx, pending, dx_root = None, None, None
        if ctx.needs_input_grad[0]:
            if gathered:
                dh = gemm_grouped_rows(dout, weight, rg.rel_ptr, trans_b=True, a_rows=rg.seg_dst,
                                       out_dtype=torch.bfloat16 if _bf16_ok(d_in) else torch.float32)
            else:
                to16 = bf16_storage() and _bf16_ok(d_in) and _xw_ok(dy, dy.size(0), d_in, d_out)
                dh = gemm_grouped_rows(dy, weight, rg.rel_ptr, trans_b=True,
                                       out_dtype=torch.bfloat16 if to16 else torch.float32)   # [S, in], a gather table
            c = rg.src_csr
            x16 = x.dtype == torch.bfloat16 and _bf16_ok(d_in)
            # [N, in]; bf16 where dx itself leaves as a bf16 table
            root16 = x16 and dout.dtype == torch.bfloat16 and _xw_ok(dout, n, d_in, d_out)
            dx_root = gemm(dout, root, trans_b=True, out_dtype=torch.bfloat16 if root16 else torch.float32)
            # x is the layer below's dropout(relu(.)) output: dx leaves through that layer's backward gate (GateLink)
            li = ctx.link_in
            gate_here = (GATE_IN_PRODUCT and li is not None and li.relu and x16 and x.is_contiguous()
                         and x.size(0) == n_table and n_table >= 4096)
            dx = spmm(c.rowptr, c.col, rg.src_w, _table(dh), n_table, addend=dx_root, split=c.split,
                      out_dtype=torch.bfloat16 if x16 else torch.float32, out_gate=x if gate_here else None,
                      out_gate_scale=li.scale if gate_here else 1.0)
            if gate_here:
                li.applied, li.db = True, None
        if branch:
            main.wait_stream(side)
        else:
            dweight, droot, db = weight_grads()
        if dx is not None and dx.dtype != x.dtype:
            dx = dx.to(x.dtype)
        return dx, dweight, droot, db, None, None, None, None, None, None, None


def rgcn_conv(x, weight, root, bias, rg, relu=False, drop=None, hidden=False, link_in=None, link_out=None):
    """``link_in`` / ``link_out`` (``GateLink``): the hand-over with the layer below / above in a stack."""
    L.require_cuda(x, weight, root, bias)
    if x.dtype != torch.bfloat16:
        x = _f32c(x)
    return _RgcnConv.apply(x, weight, root, bias, rg, relu, drop, hidden, torch.is_grad_enabled(), link_in, link_out)


# ----------------------------------------------------------------------------- composable pieces
# (used by the node-partitioned layers of biomedkg_amd/partition.py, where the gathered table is the
#  all-gathered [world * n_max, d] matrix and the output rows are the rank's own nodes)
class _SpmmFn(torch.autograd.Function):
    """``out = epi(S @ table (+ addend))`` for a rectangular sparse ``S`` given as a forward CSR (rows = outputs,
    col = table rows) and its transpose (rows = table rows, col = outputs) with matching weights."""

    @staticmethod
    def forward(ctx, table, addend, bias, fwd, bwd, w_fwd, w_bwd, n_out, relu, drop):
        out = spmm(fwd.rowptr, fwd.col, w_fwd, _table(table), n_out, addend=addend, bias=bias, relu=relu, drop=drop,
                   split=fwd.split)
        ctx.bwd, ctx.w_bwd, ctx.n_table, ctx.relu = bwd, w_bwd, table.size(0), relu
        ctx.scale = 1.0 / (1.0 - drop.p) if (drop is not None and drop.p > 0) else 1.0
        ctx.gated = relu or ctx.scale != 1.0
        ctx.keep = _plain_dropout_keep(relu, drop)
        ctx.save_for_backward(out if ctx.gated else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dout = _f32c(dout)
        if ctx.gated:
            dout = _gate(dout, out, ctx.scale) if ctx.relu else dout * ctx.keep * ctx.scale
        dtable = None
        if ctx.needs_input_grad[0]:
            dtable = spmm(ctx.bwd.rowptr, ctx.bwd.col, ctx.w_bwd, _table(dout), ctx.n_table, split=ctx.bwd.split)
        dadd = dout if ctx.needs_input_grad[1] else None
        db = colsum(dout) if ctx.needs_input_grad[2] else None
        return dtable, dadd, db, None, None, None, None, None, None, None


def spmm_fn(table, fwd, bwd, w_fwd, w_bwd, n_out, addend=None, bias=None, relu=False, drop=None):
    L.require_cuda(table, addend, bias)
    return _SpmmFn.apply(_f32c(table), addend, bias, fwd, bwd, w_fwd, w_bwd, n_out, relu, drop)


class _GroupedLinear(torch.autograd.Function):
    """``y[rows of g] = h[rows of g] @ weight[g]`` (RGCNConv's per-relation products)."""

    @staticmethod
    def forward(ctx, h, weight, group_ptr):
        ctx.group_ptr = group_ptr
        ctx.save_for_backward(h, weight)
        return gemm_grouped_rows(h, weight, group_ptr)

    @staticmethod
    def backward(ctx, dy):
        h, weight = ctx.saved_tensors
        dy = _f32c(dy)
        dh = gemm_grouped_rows(dy, weight, ctx.group_ptr, trans_b=True) if ctx.needs_input_grad[0] else None
        dw = gemm_grouped_k(h, dy, ctx.group_ptr, weight.size(0)) if ctx.needs_input_grad[1] else None
        return dh, dw, None


def grouped_linear(h, weight, group_ptr):
    L.require_cuda(h, weight, group_ptr)
    return _GroupedLinear.apply(_f32c(h), weight, group_ptr)


# ----------------------------------------------------------------------------- KGE scorer
KGE_DISTMULT, KGE_TRANSE = 0, 1
KGE_BWD_STREAMS = __import__("os").environ.get("BKG_KGE_BWD_STREAMS", "1") != "0"
KGE_LOSS_FINISH = __import__("os").environ.get("BKG_KGE_LOSS_FINISH", "1") != "0"     # 0: the loss tail as torch expressions
KGE_ONEPASS = __import__("os").environ.get("BKG_KGE_ONEPASS", "1") != "0"            # 0: the positives through scorer + list walk too
KGE_BWD_TAIL = __import__("os").environ.get("BKG_KGE_BWD_TAIL", "1") != "0"
# the fresh NEGATIVES of a whole-graph DistMult step in ONE walk over (node, role)-sorted lists, d rel_emb on the matrix cores
# (csrc/kge_onepass_mx.hip); 0: scorer + node-side list walk (the two-pass form)
KGE_ONEPASS_MX = __import__("os").environ.get("BKG_KGE_ONEPASS_MX", "1") != "0"
RGCN_KEEP_SEGMENT_MEANS = __import__("os").environ.get("BKG_RGCN_KEEP_SEGMENT_MEANS", "1") != "0"   # 0: recomputed every step           # 0: dz * gloss + the L2 term as a pass of its own


class Incidence:
    """Node -> incident (triplet, role) lists for the non-atomic scorer backward: the 2T entries
    ``cat(head, tail)`` sorted by node.  Entry e < T: the node is the head of triplet e."""

    def __init__(self, edge_index, num_nodes):
        from .graph import _Sorter
        ei = edge_index.contiguous()
        flat = ei.view(-1)                       # == cat(head, tail)
        _, self.perm, self.rowptr = _Sorter.sort_rowptr(flat, None, num_nodes, 1, num_nodes + 1)
        from .graph import rowsplit_for
        # mini-batches (a few thousand nodes, a few 10^4 incidences): rows are short, the long-row table is 6 launches
        small = num_nodes <= 4096 and flat.numel() < (1 << 18)
        self.split = None if small else rowsplit_for(self.rowptr, num_nodes, flat.numel())


class LocalIncidence:
    """The incidence lists of a node-partitioned rank's OWN nodes over the triplets it scores (``slot_index [2, T]``: table
    slots of head / tail): the entries of ``cat(head, tail)`` whose node is one of the rank's ``n_local`` rows starting at
    slot ``row0``, sorted by local row.  ``perm`` holds positions in the 2T-entry list (entry e < T: the node is the
    head of triplet e), exactly what the whole-graph ``Incidence`` holds -- so the same kernels reduce these lists into a
    ``[n_local, d]`` gradient: a triplet is scored by the owners of BOTH its endpoints, nothing is reduce-scattered.
    ``entries`` (optional, int64): the positions of the local entries when the caller knows them by construction (fresh
    negatives drawn block by block: no mask, no count read)."""

    def __init__(self, slot_index, row0, n_local, entries=None):
        from .graph import _Sorter, rowsplit_for
        flat = slot_index.contiguous().view(-1)
        if entries is None:
            entries = torch.nonzero((flat >= row0) & (flat < row0 + n_local)).squeeze(1)     # (one host read: static sets)
        if entries.numel() == flat.numel():              # every entry is local (one rank): the whole-graph incidence lists
            keys = flat if row0 == 0 else flat - row0
            _, self.perm, self.rowptr = _Sorter.sort_rowptr(keys.contiguous(), None, max(n_local, 1), 1, n_local + 1)
        else:
            keys = (flat[entries] - row0).contiguous()
            _, perm, self.rowptr = _Sorter.sort_rowptr(keys, None, max(n_local, 1), 1, n_local + 1)
            e32 = entries if entries.dtype == torch.int32 else getattr(entries, "_bkg_i32", None)
            if e32 is None:                              # (a static position list keeps its int32 form)
                e32 = entries._bkg_i32 = entries.to(torch.int32)
            self.perm = torch.empty_like(perm)
            L.call("bkg_gather_i32", L.ptr(e32), L.ptr(perm), L.ptr(self.perm), perm.numel(), L.stream())
        self.n_inc, self.n_rows, self.row0 = int(entries.numel()), int(n_local), int(row0)
        small = n_local <= 4096 and self.n_inc < (1 << 18)
        self.split = None if small else rowsplit_for(self.rowptr, n_local, self.n_inc)


class OwnedTriplets:
    """One label's triplets as a node-partitioned rank scores them (biomedkg_amd/partition.py, both-owner scoring):
    ``own`` -- the head is one of the rank's nodes: scored, counted in the loss and in d rel_emb -- then ``foreign`` --
    the tail is local, the head is not: scored for d score only (the head's owner counts them).  Indices are table slots.
    ``incidence``: the rank's nodes' lists over ``cat(own, foreign)``."""

    def __init__(self, own_index, own_type, foreign_index, foreign_type, row0, n_local, entries=None, both=None):
        """``both`` (optional): ``cat(own_index, foreign_index, dim=1)`` when the caller already holds it (the two index
        tensors are then usually its column ranges: rows contiguous, which is all the kernels need)."""
        rows_ok = lambda t: t if (t.dim() == 2 and t.stride(1) == 1) else t.contiguous()       # noqa: E731
        self.own_index, self.own_type = rows_ok(own_index), own_type.contiguous()
        self.foreign_index, self.foreign_type = rows_ok(foreign_index), foreign_type.contiguous()
        self.n_own, self.n_foreign = int(own_index.size(1)), int(foreign_index.size(1))
        self.row0, self.n_local, self._entries, self._inc = int(row0), int(n_local), entries, None
        self._both = both

    @property
    def incidence(self):
        if self._inc is None:
            both = self._both if self._both is not None else torch.cat([self.own_index, self.foreign_index], dim=1)
            self._inc = LocalIncidence(both, self.row0, self.n_local, self._entries)
            self._both = None
        return self._inc


class OnePassLists:
    """What ``bkg_kge_onepass_f32`` walks for a graph's POSITIVE triplets, built once per (edge_index, edge_type): the 2 T
    incidences ``cat(head, tail)`` sorted by (node, role, relation) -- role 0: the node is the triplet's head --, their
    streams in list order (``other``: the row to gather, ``tid``: the triplet), the work items (a node's list, hubs cut into
    slices: ``RowSplit``), the runs of equal (role, relation) inside every item cut into batches of at most ``nb`` entries,
    and the record rows: one per (head-role) run, with the static order that groups them by relation for the caller's
    segment reduce into ``d rel_emb``.  ``nb`` = ``bkg_kge_onepass_batch(d, bf16)``."""

    def __init__(self, edge_index, edge_type, num_nodes, num_relations, nb):
        from .graph import RowSplit, _Sorter
        ei = edge_index.contiguous()
        dev, t, r, n = ei.device, ei.size(1), int(num_relations), int(num_nodes)
        et = edge_type.contiguous()
        node = ei.view(-1)                                               # cat(head, tail)
        lo = torch.cat([et, et + r])                                     # role * R + relation
        keys, perm, rowptr = _Sorter.sort_rowptr(lo, node, 2 * r, n, n + 1, 2 * r)
        perm64 = perm.long()
        hd = perm64 < t
        trip = torch.where(hd, perm64, perm64 - t)
        self.other = torch.where(hd, ei[1][trip], ei[0][trip]).to(torch.int32).contiguous()
        self.tid = trip.to(torch.int32).contiguous()
        relrole = (keys.long() % (2 * r))                                # per entry, list order
        self.split = RowSplit(rowptr, n, 2 * t)
        self.split.tighten()                                             # (one host read per graph: the item count)
        n_items = self.split.n_items_max
        item_row, item_beg, item_end, item_slot = (b[:n_items].long() for b in self.split.bufs[:4])
        # runs: a new one where an item starts or the (role, relation) key changes
        start = torch.zeros(2 * t + 1, dtype=torch.bool, device=dev)
        start[item_beg[item_end > item_beg]] = True
        if t:
            start[1:2 * t] |= relrole[1:] != relrole[:-1]
        run_first = torch.nonzero(start[:2 * t]).squeeze(1)
        n_runs = run_first.numel()
        run_len = torch.diff(run_first, append=torch.tensor([2 * t], device=dev))
        run_key = relrole[run_first]
        nbr = (run_len + nb - 1) // nb                                   # batches per run
        n_b = int(nbr.sum())
        b_run = torch.repeat_interleave(torch.arange(n_runs, device=dev), nbr)
        b_k = torch.arange(n_b, device=dev) - (torch.cumsum(nbr, 0) - nbr)[b_run]
        b_e0 = run_first[b_run] + b_k * nb
        b_cnt = torch.minimum(torch.full_like(b_k, nb), run_len[b_run] - b_k * nb)
        head_run = run_key < r
        rec_of_run = torch.cumsum(head_run.long(), 0) - 1
        self.n_records = int(head_run.sum())
        flags = (b_k == 0).long() * 0x100 + (b_k == nbr[b_run] - 1).long() * 0x200
        b_rec = torch.where(head_run[b_run], rec_of_run[b_run], torch.full_like(b_run, -1))
        self.batches = torch.stack([b_e0, b_cnt + flags, run_key[b_run], b_rec], dim=1).to(torch.int32).contiguous()
        i_b0 = torch.searchsorted(b_e0, item_beg)
        i_b1 = torch.searchsorted(b_e0, item_end)
        self.items = torch.stack([item_row, i_b0, i_b1, item_slot], dim=1).to(torch.int32).contiguous()
        self.n_items, self.n_batches, self.nb = n_items, n_b, nb
        # the records grouped by relation (stable): rows of the segment reduce that sums them into d rel_emb
        rec_rel = run_key[head_run]
        _, self.rec_order, self.rec_ptr = _Sorter.sort_rowptr(rec_rel.contiguous(), None, r, 1, r + 1)
        from .graph import rowsplit_for
        self.rec_split = rowsplit_for(self.rec_ptr, r, self.n_records)


def onepass_lists_of(edge_index, edge_type, num_nodes, num_relations, nb) -> OnePassLists:
    """Cached on ``edge_index`` for THIS ``edge_type`` object at this version (an id tensor re-allocated at the same
    address -- ``fix_edge_id``'s ``full_like`` every step -- is another object: ``onepass_stable`` keeps such flows off
    this path, and a stale entry is rebuilt here)."""
    from .graph import _cached
    key = (int(num_nodes), int(num_relations), int(nb), edge_type.data_ptr(), edge_type._version)

    def build():
        lists = OnePassLists(edge_index, edge_type, num_nodes, num_relations, nb)
        lists.types_ref = weakref.ref(edge_type)
        return lists
    lists = _cached(edge_index, "onepass", key, build)
    if lists.types_ref() is not edge_type:
        edge_index._bkg_cache.pop("onepass", None)
        lists = _cached(edge_index, "onepass", key, build)
    return lists


def onepass_stable(edge_index, edge_type):
    """Has this very (``edge_index``, ``edge_type``) pair of tensor objects, at these versions, been asked about before?
    The lists of the one-pass walk take ~20 launches over 8 M entries to build: they pay from the second step of a graph
    that stays, never for triplets that are new objects every step (mini-batches, ``fix_edge_id``)."""
    seen = getattr(edge_index, "_bkg_onepass_seen", None)
    now = (edge_index._version, edge_type._version)
    same = seen is not None and seen[0]() is edge_type and seen[1] == now
    if not same:
        edge_index._bkg_onepass_seen = (weakref.ref(edge_type), now)
    return same


def incidence_of(edge_index, num_nodes) -> Incidence:
    from .graph import _cached, capturing

    def build():
        inc = Incidence(edge_index, num_nodes)
        # a structure known to stay (the sorted frame of a full graph's positives, built on the graph's second step) takes
        # its one host read NOW: deferred to its third launch it landed in the first timed step of a three-step warm-up
        # (a drained queue: 11.2 against 9.75 ms)
        if getattr(edge_index, "_bkg_stays", False) and inc.split is not None and not capturing():
            inc.split.tighten()
        return inc
    return _cached(edge_index, "incidence", (int(num_nodes),), build)


def relation_order_of(edge_type, num_relations):
    """Triplet ids grouped by relation (stable) + per-relation offsets, cached on ``edge_type``."""
    from .graph import _cached, _Sorter

    def build():
        sorted_keys, perm, rel_ptr = _Sorter.sort_rowptr(edge_type, None, num_relations, 1, num_relations + 1)
        return perm, rel_ptr
    tag = getattr(edge_type, "_bkg_cache", None)
    if tag is not None:
        key = (edge_type._version, int(num_relations))
        own, planted = tag.get("rel_order"), tag.get("rel_order_planted")
        if (own is None or own[0] != key) and planted is not None and planted[0] == key and planted[1][2] < _PLANTED_ORDER_USES:
            planted[1][2] += 1           # the R-GCN structure's (relation, target) sort, while the structure is young
            return planted[1][0], planted[1][1]
    return _cached(edge_type, "rel_order", (int(num_relations),), build)


_PLANTED_ORDER_USES = 2     # steps a graph's triplets are walked in the R-GCN structure's order before they get their own


SORTED_POSITIVES = __import__("os").environ.get("BKG_SORTED_POSITIVES", "1") != "0"
_SORTED_MIN_TRIPLETS = 1 << 20


# how the positives' scores leave a call that ran in the sorted frame: put back in the caller's order by the op (default), or
# left in frame order with the inverse permutation handed over beside them ("lazy": a training step stores the scores for
# whoever may ask -- metrics, tests -- and pays the 4 M-float gather only then)
_FRAME = {"lazy": False, "inverse": None, "lazy_neg": None}


def scores_in_order(scores, inverse):
    """Scores of a call in the sorted frame (``_SortedTriplets``) in the caller's triplet order."""
    if inverse is None:
        return scores
    back = torch.empty_like(scores)
    L.call("bkg_gather_f32", L.ptr(scores), L.ptr(inverse), L.ptr(back), back.numel(), L.stream())
    return back


class _SortedTriplets:
    """A graph's positives stored in (relation, target) order -- the R-GCN structure's sort (graph.RelGraph) -- for the
    scorer of a graph that STAYS (full-graph training): the ids are read in place, the tails of consecutive triplets are
    the same or neighbouring rows of ``z`` (every tail row leaves the L2 once per (relation, target) segment instead of
    once per triplet), scores and gradient records are written front to back, and the tail half of every node's incidence
    list is a handful of runs of consecutive records.  ``index`` / ``types``: the triplets in that order (their relation
    order planted as the identity); ``inverse``: position of the caller's triplet in it (scores go back through it)."""

    __slots__ = ("index", "types", "inverse")

    def __init__(self, edge_index, edge_type, perm, rel_ptr, num_relations):
        p64 = perm.long()
        self.index = edge_index.index_select(1, p64).contiguous()
        self.types = edge_type.index_select(0, p64).contiguous()
        self.inverse = torch.empty_like(perm)                                  # int32 (t < 2^31)
        self.inverse[p64] = _iota32(p64.numel(), p64.device)
        self.types._bkg_cache = {"rel_order": ((self.types._version, int(num_relations)),
                                               (_iota32(p64.numel(), p64.device), rel_ptr))}
        self.index._bkg_stays = True      # (its incidence lists read their counts when built, not on their third launch)


def sorted_positives_of(edge_index, edge_type, num_relations):
    """``_SortedTriplets`` of a triplet set whose R-GCN structure has outlived its young steps (``relation_order_of``: the
    planted order was used up), else None.  Cached on ``edge_type`` (keyed on the index tensor and both versions)."""
    from .graph import _cached
    t = edge_type.numel()
    if not SORTED_POSITIVES or t < _SORTED_MIN_TRIPLETS or t >= (1 << 31):
        return None
    tag = getattr(edge_type, "_bkg_cache", None)
    planted = tag.get("rel_order_planted") if tag is not None else None
    key = (edge_type._version, int(num_relations))
    if planted is None or planted[0] != key or planted[1][2] < _PLANTED_ORDER_USES:
        return None
    perm, rel_ptr = planted[1][0], planted[1][1]
    if perm.numel() != t or edge_index.size(1) != t:
        return None
    return _cached(edge_type, "sorted_triplets", (int(num_relations), edge_index.data_ptr(), edge_index._version),
                   lambda: _SortedTriplets(edge_index, edge_type, perm, rel_ptr, num_relations))


def grouped_types(edge_type, repeat, num_relations):
    """``edge_type.repeat(repeat)[randperm(...)]`` (reference kge_module.py:97-100) delivered GROUPED BY RELATION: the
    multiset of the positives' relation ids, ``repeat`` times, in relation order.  The negatives it is paired with come out of
    the sampler in exchangeable order (the first K survivors of a keyed random permutation of the candidate range,
    sampling.negative_sampling), so ANY fixed arrangement of the ids is a uniformly random pairing -- the set of typed
    negatives has the reference's distribution -- and this one is the arrangement the fused scorer wants: its relation
    order is the identity (no permutation launch, no sort, head / tail ids read in place).  Cached on ``edge_type``
    together with its own relation order."""
    from .graph import _cached
    t = edge_type.numel() * int(repeat)
    if edge_type.dim() != 1 or edge_type.dtype != torch.int64 or not edge_type.is_cuda or t == 0 or t >= (1 << 31):
        return None

    def build():
        _, rel_ptr = relation_order_of(edge_type, num_relations)
        ptr = rel_ptr * int(repeat)                                      # (int32; t < 2^31)
        iota = _iota32(t, edge_type.device)
        # id of position i = number of group ends <= i (one launch; torch.repeat_interleave walks each group serially)
        types = torch.searchsorted(ptr[1:], iota, right=True)
        types._bkg_cache = {"rel_order": ((types._version, int(num_relations)), (iota, ptr))}
        return types
    return _cached(edge_type, "grouped_types", (int(repeat), int(num_relations)), build)


class _KgeScore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, rel_emb, edge_index, edge_type, kind):
        t = edge_index.size(1)
        score = torch.empty(t, dtype=torch.float32, device=z.device)
        L.call("bkg_kge_score_f32", kind, L.ptr(z), z.stride(0), z.size(1), L.ptr(edge_index[0]), L.ptr(edge_index[1]),
               L.ptr(edge_type), t, L.ptr(rel_emb), rel_emb.stride(0), L.ptr(score), L.stream())
        ctx.kind = kind
        ctx.edge_index, ctx.edge_type = edge_index, edge_type
        ctx.save_for_backward(z, rel_emb)
        return score

    @staticmethod
    def backward(ctx, gscore):
        z, rel_emb = ctx.saved_tensors
        ei, et, kind = ctx.edge_index, ctx.edge_type, ctx.kind
        gscore = _f32c(gscore)
        n, d, t = z.size(0), z.size(1), ei.size(1)
        dz = drel = None
        if ctx.needs_input_grad[0]:
            inc = incidence_of(ei, n)
            l1 = None
            if kind == KGE_TRANSE:
                l1 = torch.empty(n, dtype=torch.float32, device=z.device)
                L.call("bkg_row_l1norm_f32", L.ptr(z), z.stride(0), n, d, L.ptr(l1), L.stream())
            dz = torch.empty_like(z)
            sp = inc.split.use() if inc.split is not None else None
            partials = sp.partials(d, z.device) if sp is not None else None
            L.call("bkg_kge_bwd_nodes_f32", kind, L.ptr(z), 0, z.stride(0), n, d, L.ptr(ei[0]), L.ptr(ei[1]), L.ptr(et), t,
                   L.ptr(rel_emb), rel_emb.stride(0), rel_emb.size(0), L.ptr(gscore), None, L.ptr(l1), L.ptr(inc.rowptr),
                   L.ptr(inc.perm),
                   L.ptr(dz), dz.stride(0), L.sref(sp.c if sp is not None else None), L.ptr(partials), 0, 0, L.stream())
        if ctx.needs_input_grad[1]:
            drel = torch.zeros_like(rel_emb)
            order, rel_ptr = relation_order_of(et, rel_emb.size(0)) if t >= (1 << 15) else (None, None)
            L.call("bkg_kge_bwd_rel_f32", kind, L.ptr(z), z.stride(0), d, L.ptr(ei[0]), L.ptr(ei[1]), L.ptr(et), t,
                   L.ptr(rel_emb), rel_emb.stride(0), L.ptr(gscore), rel_emb.size(0), L.ptr(drel), drel.stride(0),
                   L.ptr(order), L.ptr(rel_ptr), L.stream())
        return dz, drel, None, None, None


def kge_score(z, rel_emb, edge_index, edge_type, kind):
    """DistMult / TransE logits of the triplets ``(edge_index[0], edge_type, edge_index[1])``."""
    L.require_cuda(z, rel_emb, edge_index, edge_type)
    if edge_index.dtype != torch.int64 or edge_type.dtype != torch.int64:
        raise TypeError("edge_index / edge_type must be int64")
    if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_type.numel() != edge_index.size(1):
        raise ValueError(f"edge_index {tuple(edge_index.shape)} and edge_type {tuple(edge_type.shape)} do not match")
    return _KgeScore.apply(_f32c(z), _f32c(rel_emb), edge_index.contiguous(), edge_type.contiguous(), kind)


_KGE_PIECES = (2, 4, 6, 8, 12, 16, 24, 32, 64, 96, 128, 192)   # csrc/rowpiece.h BKG_PIECE_DISPATCH
class _FusedChunk:
    """Triplets per wave of csrc/kge_fused.hip, asked of the library at first use."""
    value = None

    def __int__(self):
        if _FusedChunk.value is None:
            _FusedChunk.value = int(L.lib().bkg_kge_fused_chunk())
        return _FusedChunk.value


_FUSED_CHUNK_OF = _FusedChunk()


_FEW_WAVES = 1 << 11     # below this the per-relation reduce of the waves' rows needs no long-row table (one wave then sums a
                         # relation's rows in turn: ~70 at 30 relations; a node-partitioned rank's share -- 8k waves at 8 ranks --
                         # took 0.34 ms per triplet set that way, r05 simulated-rank trace, and goes through the row split now)


def _iota32(n, device):
    from .graph import iota
    return iota(int(n), torch.int32, device)


def _relation_waves_small(rel_ptr, num_relations, n_waves):
    """The same wave CSR for a mini-batch's few waves: one launch, no long-row table, nothing cached."""
    wave_ptr = torch.empty(num_relations + 1, dtype=torch.int32, device=rel_ptr.device)
    L.call("bkg_relation_wave_ptr", L.ptr(rel_ptr), num_relations, int(_FUSED_CHUNK_OF), L.ptr(wave_ptr), L.stream())
    return wave_ptr, _iota32(n_waves, rel_ptr.device), None


def _relation_waves(edge_type, rel_ptr, num_relations, n_waves):
    """Wave ranges of each relation in the fused pass (relation r owns ceil(cnt_r / 64) consecutive waves) as a
    CSR over wave rows + its long-row table, cached with the relation order on ``edge_type``."""
    from .graph import RowSplit, _cached

    def build():
        cnt = (rel_ptr[1:] - rel_ptr[:-1]).long()
        ch = int(_FUSED_CHUNK_OF)
        nw = (cnt + ch - 1) // ch
        wave_ptr = torch.cat([nw.new_zeros(1), torch.cumsum(nw, 0)]).to(torch.int32)
        iota = torch.arange(n_waves, dtype=torch.int32, device=edge_type.device)
        return wave_ptr, iota, RowSplit(wave_ptr, num_relations, n_waves)
    return _cached(edge_type, "rel_waves", (int(num_relations), int(n_waves)), build)


def kge_prepare_structures(pos_edge_index, pos_edge_type, neg_edge_index, neg_edge_type, num_nodes, num_groups):
    """Build (and cache on the tensors) what ``kge_bce_loss`` and its backward look up for these triplets -- the
    relation orders and the node incidence lists -- ahead of time, e.g. on a side stream while the encoder runs."""
    for ei, et in ((pos_edge_index, pos_edge_type), (neg_edge_index, neg_edge_type)):
        if ei.size(1):
            relation_order_of(et, num_groups)
            incidence_of(ei, num_nodes)


def role_incidence(edge_index, edge_type, num_nodes):
    """-> (rowptr2 int32 [2 N + 1], packed int32 [2 T]): the incidences cat(head, tail) of a triplet set sorted by
    (node, role) -- role 0: the node is the head -- each as (other endpoint | relation << 24): what the one-pass walk of the
    negatives reads (csrc/graph.hip bkg_kge_role_incidence: one key-making launch, one radix sort of pairs, one lower bound;
    no host read)."""
    t, dev = edge_index.size(1), edge_index.device
    ws_bytes = L.lib().bkg_sort_workspace_bytes(max(2 * t, 1))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    keys, vals, skeys, packed = (torch.empty(2 * t, dtype=torch.int32, device=dev) for _ in range(4))
    rowptr2 = torch.empty(2 * num_nodes + 1, dtype=torch.int32, device=dev)
    L.call("bkg_kge_role_incidence", L.ptr(edge_index[0]), L.ptr(edge_index[1]), L.ptr(edge_type), t, num_nodes, L.ptr(keys),
           L.ptr(vals), L.ptr(skeys), L.ptr(packed), L.ptr(rowptr2), L.ptr(ws), ws_bytes, L.stream())
    return rowptr2, packed


class LazyScores:
    """The negatives' scores of a training step whose one-pass walk did not write them: computed by the plain scorer from
    the step's own table when somebody asks (``KGEModule._last_scores``)."""

    def __init__(self, table, rel_emb, edge_index, edge_type, kind):
        self.args = (table, rel_emb.detach(), edge_index, edge_type, kind)
        self.value = None

    def __call__(self):
        if self.value is None:
            table, rel_emb, ei, et, kind = self.args
            with torch.no_grad():
                self.value = kge_score(table.float() if table.dtype != torch.float32 else table, rel_emb, ei, et, kind)
            self.args = None
        return self.value


_FINISH_SCRATCH = {}


def _finish_scratch(device):
    """Slots + ticket of bkg_kge_loss_finish_f32 (zeroed once; launches on one stream at a time use it in turn)."""
    key = str(device)
    if key not in _FINISH_SCRATCH:
        _FINISH_SCRATCH[key] = torch.zeros(L.lib().bkg_kge_loss_finish_scratch_bytes(), dtype=torch.uint8, device=device)
    return _FINISH_SCRATCH[key]


class _KgeBceLoss(torch.autograd.Function):
    """``binary_cross_entropy_with_logits(cat(decode(pos), decode(neg)), cat(1, 0))`` of the reference's
    ``KGEModule.training_step`` (kge_module.py:121-127) as ONE op: the forward sweeps each triplet set once
    (scores, loss terms, d loss / d score, d rel_emb -- csrc/kge_fused.hip), the backward is the per-node
    segment reduce for dz.  Returns ``(loss, pos_pred, neg_pred)``; the preds are detached outputs.

    ``owned`` (node-partitioned training, biomedkg_amd/partition.py): ``(OwnedTriplets of the positives, of the
    negatives)`` in table slots -- the rank scores every triplet that touches one of its nodes; the ones whose HEAD it
    owns count in the loss and in d rel_emb, the others (tail local, head remote) are scored for d score alone, and dz is
    reduced over the rank's own nodes' lists only: ``[n_local, d]``, no reduce-scatter."""

    @staticmethod
    def forward(ctx, z, rel_emb, pos_ei, pos_et, neg_ei, neg_et, kind, comm, total, reg, inv_total, rows, reg_rows=None,
                owned=None):
        d, r = z.size(1), rel_emb.size(0)
        # padded mini-batches (biomedkg_amd/sampled.py): ``inv_total`` is a DEVICE scalar, pad triplets carry the
        # relation id ``r`` -- one group past the real ones, which no wave visits -- and their records stay zero
        padded = inv_total is not None
        dev = z.device
        # bf16 mode: both passes gather the node rows from a bf16 copy of z (half the bytes; fp32 accumulate)
        zt = to_bf16(z) if (bf16_storage() and d % 8 == 0 and (d // 8) in _KGE_PIECES) else z
        if comm is not None:              # node-partitioned: the scorer reads the all-gathered table (triplet ids = slots)
            if owned is None:
                raise ValueError("a node-partitioned scorer takes its triplets as OwnedTriplets (kge_bce_loss_owned)")
            zt = comm.all_gather(zt)
        n = zt.size(0)
        zt_bf16 = int(zt.dtype == torch.bfloat16)
        # triplet sets (index, type, label, counted) in groups: one group = one incidence structure of the backward
        resort = None
        # A whole graph's POSITIVES in a DistMult training pass: ONE walk over their (node, role, relation)-sorted lists does
        # scorer and gradient (csrc/kge_bwd_nodes.hip kge_onepass_kernel; lists built once per graph) -- dz's positive part
        # is ready when the forward returns, the backward adds the negatives' part to it
        onepass = (KGE_ONEPASS and owned is None and not padded and comm is None and kind == KGE_DISTMULT
                   and bool(ctx.needs_input_grad[0]) and pos_ei.size(1) >= _SORTED_MIN_TRIPLETS and 2 * pos_ei.size(1) < (1 << 31)
                   and n < (1 << 31) and 0 < L.lib().bkg_kge_onepass_smem_bytes(d, zt_bf16, r) <= 65536
                   and zt.stride(0) % (8 if zt_bf16 else 4) == 0
                   and rel_emb.stride(0) % 4 == 0 and rel_emb.data_ptr() % 16 == 0 and not torch.cuda.is_current_stream_capturing()
                   and onepass_stable(pos_ei, pos_et))
        # ... and then the step's NEGATIVES too, in a walk of their own over (node, role)-sorted lists built this step (2 T row
        # reads instead of the scorer's 2 T + the list walk's 2 T; d rel_emb as a one-hot product on the matrix cores).  Their
        # scores are not written (a training step never reads them): ``LazyScores`` computes them if somebody asks
        # Only for negatives that come out of the library's uniform sampler (it marks its output ``_bkg_exchangeable``: a
        # uniform subset of the free cells): the walk gives a node's whole list to ONE lane group -- lists of Poisson length
        # -- where the two-pass form cuts a hub's list into slices
        mx = (KGE_ONEPASS_MX and onepass and _FRAME["lazy"] and zt_bf16 and neg_ei.size(1) >= _SORTED_MIN_TRIPLETS
              and getattr(neg_ei, "_bkg_exchangeable", False)
              and 2 * neg_ei.size(1) < (1 << 31) and n <= (1 << 24) and n * zt.stride(0) < (1 << 31) and r < 256
              and zt.stride(0) % 8 == 0
              and bool(L.lib().bkg_kge_onepass_mx_ok(d, 1, r)))
        pos_one = None
        if owned is None:
            if not padded and comm is None and not onepass:
                resort = sorted_positives_of(pos_ei, pos_et, r)
                if resort is not None:             # the positives of a graph that stays: walked in (relation, target) order
                    pos_ei, pos_et = resort.index, resort.types
            sets = ((pos_ei, pos_et, 1.0, True, 0), (neg_ei, neg_et, 0.0, True, 1))
            if onepass:
                sets = () if mx else sets[1:]
        else:
            po, ne = owned
            sets = ((po.own_index, po.own_type, 1.0, True, 0), (po.foreign_index, po.foreign_type, 1.0, False, 0),
                    (ne.own_index, ne.own_type, 0.0, True, 1), (ne.foreign_index, ne.foreign_type, 0.0, False, 1))
        t_all = sum(ei.size(1) for ei, _, _, _, _ in sets)
        if total is None:
            total = (sum(ei.size(1) for ei, _, _, counted, _ in sets if counted) + (pos_ei.size(1) if onepass else 0)
                     + (neg_ei.size(1) if mx else 0))
        # packed {head, tail, rel, g} per triplet for the node-side backward (ids fit int32; one gather instead of four)
        pack = bool(ctx.needs_input_grad[0]) and n < (1 << 31) and r < (1 << 31)
        if padded and not pack:
            raise ValueError("padded triplets need the packed gradient records (z must require grad)")
        if owned is not None and ctx.needs_input_grad[0] and not pack:
            raise ValueError("both-owner scoring needs the packed gradient records (table rows and relations < 2^31)")
        n_waves = [L.lib().bkg_kge_fused_bce_waves(ei.size(1), r) if ei.size(1) else 0 for ei, _, _, _, _ in sets]
        lists, one_parts, mx_parts, mx_blocks = None, 0, 0, 0
        if onepass:
            lists = onepass_lists_of(pos_ei, pos_et, n, r, L.lib().bkg_kge_onepass_batch(d, zt_bf16))
            one_parts = 4 * L.lib().bkg_kge_onepass_blocks(lists.n_items)
            if mx:
                mx_blocks = L.lib().bkg_kge_onepass_mx_blocks(n)
                mx_parts = 4 * mx_blocks
                one_parts += mx_parts                    # (the negatives' walk takes the slots behind the positives')
        part = torch.empty(one_parts + sum(n_waves), dtype=torch.float32, device=dev)      # every wave of a launch writes its slot
        # the sets the loss counts take the first slots: ONE contiguous sum (the others' slots follow)
        n_counted = one_parts + sum(w for w, st in zip(n_waves, sets) if st[3])
        slot_c, slot_u, slot0 = one_parts, n_counted, []
        for w, st in zip(n_waves, sets):
            slot0.append(slot_c if st[3] else slot_u)
            if st[3]:
                slot_c += w
            else:
                slot_u += w
        new = torch.zeros if padded else torch.empty
        rec_all = new(t_all, 4, dtype=torch.int32, device=dev) if pack else None
        scores, grads, drel = [], [], None
        t0 = 0
        dz_neg = None
        if mx:
            t_neg = neg_ei.size(1)
            rowptr2, packed = role_incidence(neg_ei, neg_et, n)
            dz_neg = torch.empty(n, d, dtype=torch.float32, device=dev)
            drel_mx = torch.empty(mx_blocks, 32 * d, dtype=torch.float32, device=dev)
            # algorithmic bytes: per incidence the other endpoint's row + its packed index; per node its own row, the
            # gradient row and its two list bounds; the workgroups' partial sums of d rel_emb
            with _Timed("kge_onepass_mx", 2 * t_neg * (d * zt.element_size() + 4) + n * (d * zt.element_size() + 4 * d + 8)
                        + mx_blocks * 32 * d * 4, d=d, launches_per_call=1):
                L.call("bkg_kge_onepass_mx_f32", L.ptr(zt), zt.stride(0), n, L.ptr(rel_emb), rel_emb.stride(0), r,
                       L.ptr(rowptr2), L.ptr(packed), None, 2 * t_neg, 0.0, 1.0 / total, None,
                       L.ptr(part[one_parts - mx_parts:one_parts]), L.ptr(dz_neg), dz_neg.stride(0), L.ptr(drel_mx),
                       mx_blocks, L.stream())
            drel = (drel_mx.sum(0).view(32, d)[:r] * (1.0 / total)).contiguous()
        if onepass:
            t_pos = pos_ei.size(1)
            s_pos = torch.empty(t_pos, dtype=torch.float32, device=dev)
            dz_pos = torch.empty(n, d, dtype=torch.float32, device=dev)
            records = torch.empty(lists.n_records, d, dtype=torch.float32, device=dev)
            sp = lists.split.use()
            # algorithmic bytes: per incidence the other endpoint's row + its index and triplet id; per node its own row, the
            # gradient row and its item; per run a record row written and read
            with _Timed("kge_onepass", 2 * t_pos * (d * zt.element_size() + 8) + n * (d * zt.element_size() + 4 * d + 16)
                        + 2 * lists.n_records * 4 * d + lists.n_batches * 16, d=d, launches_per_call=1):
                L.call("bkg_kge_onepass_f32", L.ptr(zt), zt_bf16, zt.stride(0), n, d, L.ptr(rel_emb), rel_emb.stride(0), r,
                       L.ptr(lists.other), L.ptr(lists.tid), L.ptr(lists.batches), L.ptr(lists.items), lists.n_items, 1.0,
                       1.0 / total, L.ptr(s_pos), L.ptr(part[:one_parts - mx_parts]), L.ptr(records), L.ptr(dz_pos),
                       dz_pos.stride(0), L.sref(sp.c), L.ptr(sp.partials(d, dev)), L.stream())
            # the runs' record rows added per relation (fixed order: the records are grouped once per graph)
            dr = spmm(lists.rec_ptr, lists.rec_order, None, records, r, split=lists.rec_split)
            drel = dr if drel is None else drel + dr
            pos_one = (s_pos, dz_pos)
        for (ei, et, label, counted, _), waves, w0 in zip(sets, n_waves, slot0):
            t = ei.size(1)
            s = torch.empty(t, dtype=torch.float32, device=dev)
            rec = rec_all[t0:t0 + t] if (pack and t) else None
            g = torch.empty(t if rec is None else 0, dtype=torch.float32, device=dev)   # g travels inside rec when packed
            if t:
                order, rel_ptr = relation_order_of(et, r + 1 if padded else r)
                if padded:
                    rel_ptr = rel_ptr[: r + 1]
                drel_part = torch.empty(waves, d, dtype=torch.float32, device=dev)
                if waves <= _FEW_WAVES or padded:      # (a padded mini-batch: fresh ids every batch, nothing to cache a table on)
                    wave_ptr, wave_iota, wave_split = _relation_waves_small(rel_ptr, r, waves)
                else:
                    wave_ptr, wave_iota, wave_split = _relation_waves(et, rel_ptr, r, waves)
                # algorithmic bytes: two node rows + head / tail ids + order + score + the 16-byte record
                with _Timed("kge_fused_bce" if not padded else None, t * (2 * d * zt.element_size() + 16 + 4 + 4 + 16),
                            d=d, launches_per_call=1):
                    L.call("bkg_kge_fused_bce_dev_f32", kind, L.ptr(zt), zt_bf16, zt.stride(0), d, L.ptr(ei[0]), L.ptr(ei[1]),
                           t, label, 1.0 / total, L.ptr(inv_total), L.ptr(rel_emb), rel_emb.stride(0), r, L.ptr(order),
                           L.ptr(rel_ptr), L.ptr(s), L.ptr(g) if rec is None else None, L.ptr(rec),
                           L.ptr(part[w0:w0 + waves]), L.ptr(drel_part), drel_part.stride(0), L.stream())
                if counted:
                    # per-relation sum of the waves' partial rows (consecutive waves; fixed order; no atomics)
                    dr = spmm(wave_ptr, wave_iota, None, drel_part, r, split=wave_split)
                    drel = dr if drel is None else drel + dr
            scores.append(s)
            grads.append(g)
            t0 += t
        if drel is None:
            drel = torch.zeros_like(rel_emb)
        ctx.kind, ctx.comm, ctx.reg = kind, comm, float(reg)
        ctx.reg_rows = reg_rows
        ctx.owned = owned
        ctx.n_real = rows
        ctx.padded = padded
        ctx.onepass = onepass
        ctx.mx = mx
        if owned is None and onepass and mx:
            ctx.graphs = None
            ctx.save_for_backward(zt, rel_emb, None, dz_neg, drel, None, None, z if reg else None, pos_one[1])
            pos_scores = pos_one[0]
            neg_scores = torch.empty(0, dtype=torch.float32, device=dev)      # (a placeholder: see _FRAME["lazy_neg"])
            _FRAME["inverse"] = None
            _FRAME["lazy_neg"] = LazyScores(zt, rel_emb, neg_ei, neg_et, kind)
        elif owned is None and onepass:
            ctx.graphs = (pos_ei, pos_et, neg_ei, neg_et)
            ctx.save_for_backward(zt, rel_emb, None, grads[0], drel, None, rec_all if (pack and neg_ei.size(1)) else None,
                                  z if (reg or padded) else None, pos_one[1])
            pos_scores, neg_scores = pos_one[0], scores[0]
            _FRAME["inverse"] = None
        elif owned is None:
            ctx.graphs = (pos_ei, pos_et, neg_ei, neg_et)
            ctx.save_for_backward(zt, rel_emb, grads[0], grads[1], drel,
                                  rec_all[:pos_ei.size(1)] if (pack and pos_ei.size(1)) else None,
                                  rec_all[pos_ei.size(1):] if (pack and neg_ei.size(1)) else None,
                                  z if (reg or padded) else None, None)
            pos_scores, neg_scores = scores[0], scores[1]
            _FRAME["inverse"] = None
            if resort is not None and _FRAME["lazy"]:
                _FRAME["inverse"] = resort.inverse            # the caller puts them in order if anybody asks (scores_in_order)
            elif resort is not None:
                pos_scores = scores_in_order(pos_scores, resort.inverse)
        else:
            t_pos = owned[0].n_own + owned[0].n_foreign
            ctx.graphs = None
            ctx.save_for_backward(zt, rel_emb, None, None, drel, rec_all[:t_pos] if (pack and t_pos) else None,
                                  rec_all[t_pos:] if (pack and t_all > t_pos) else None, z if reg else None, None)
            pos_scores, neg_scores = scores[0], scores[2]           # (the sets the rank counts)
        ctx.mark_non_differentiable(pos_scores, neg_scores)
        if padded:   # BCE mean + L2 over the real rows in ONE launch (a captured mini-batch step is bound by its launch count)
            out = torch.empty((), dtype=torch.float32, device=dev)
            L.call("bkg_kge_loss_finish_f32", L.ptr(part), part.numel(), 0.0, L.ptr(inv_total), L.ptr(z), z.stride(0),
                   z.size(0), d, L.ptr(rows), 0, L.ptr(rel_emb), rel_emb.stride(0), r, float(reg),
                   L.ptr(_finish_scratch(dev)), L.ptr(out), L.stream())
            return out, pos_scores, neg_scores
        if (KGE_LOSS_FINISH and total > 0 and d % 4 == 0 and z.stride(0) % 4 == 0 and rel_emb.stride(0) % 4 == 0 and z.stride(1) == 1
                and rel_emb.stride(1) == 1 and z.data_ptr() % 16 == 0 and rel_emb.data_ptr() % 16 == 0 and z.size(0) < (1 << 31)):
            # BCE mean + the L2 term in ONE launch (fp64 sums in a fixed order), as the padded mini-batch steps do
            out = torch.empty((), dtype=torch.float32, device=dev)
            L.call("bkg_kge_loss_finish_f32", L.ptr(part), n_counted, 1.0 / total, None, L.ptr(z), z.stride(0), z.size(0), d,
                   None, 0.0 if reg_rows is None else float(reg_rows), L.ptr(rel_emb), rel_emb.stride(0), r, float(reg),
                   L.ptr(_finish_scratch(dev)), L.ptr(out), L.stream())
            return out, pos_scores, neg_scores
        loss = part[:n_counted].sum(dtype=torch.float64) / total
        if reg:      # + reg * (mean(z^2) + mean(rel_emb^2)), reference kge_module.py:128-129: two one-launch norms
            z_rows = z.size(0) if reg_rows is None else reg_rows        # a partitioned step: the mean over N / world rows
            loss = loss + reg * (torch.linalg.vector_norm(z).double().pow(2) / (z_rows * z.size(1))
                                 + torch.linalg.vector_norm(rel_emb).double().pow(2) / rel_emb.numel())
        return loss.float(), pos_scores, neg_scores

    @staticmethod
    def backward(ctx, gloss, _gp, _gn):
        zt, rel_emb, g_pos, g_neg, drel, rec_pos, rec_neg, z_reg, dz_pos = ctx.saved_tensors
        kind, owned = ctx.kind, ctx.owned
        n, d = zt.size(0), zt.size(1)
        zt_bf16 = int(zt.dtype == torch.bfloat16)
        z = zt
        dz = None
        if ctx.needs_input_grad[0] and getattr(ctx, "mx", False):
            # both walks ran in the forward: dz_pos holds the sums of the positives' lists, g_neg (the slot of the negatives'
            # d score, which this form never materialises) those of the negatives'.  What is left is their sum, the upstream
            # gradient and the L2 term: one pass, out of place (the buffers stay intact for a second backward)
            dz_neg = g_neg
            ok = (gloss.numel() == 1 and d % 4 == 0 and dz_pos.stride(0) % 4 == 0 and dz_neg.stride(0) % 4 == 0
                  and (not ctx.reg or (z_reg is not None and z_reg.dtype == torch.float32 and z_reg.shape == dz_pos.shape
                                       and z_reg.stride(1) == 1 and z_reg.stride(0) % 4 == 0 and z_reg.data_ptr() % 16 == 0)))
            if ok:
                dz = torch.empty_like(dz_pos)
                gl = gloss.reshape(1).float().contiguous()
                zr = z_reg if ctx.reg else None
                L.call("bkg_kge_dz_finish_f32", L.ptr(dz_pos), dz_pos.stride(0), L.ptr(dz_neg), dz_neg.stride(0), L.ptr(zr),
                       zr.stride(0) if zr is not None else 0, n, d, L.ptr(gl), 2.0 * ctx.reg / d, L.ptr(dz), dz.stride(0),
                       L.stream())
            else:
                dz = (dz_pos + dz_neg) * gloss
                if ctx.reg:
                    dz = torch.addcmul(dz, z_reg, gloss * (2.0 * ctx.reg / (z_reg.size(0) * z_reg.size(1))))
        elif ctx.needs_input_grad[0]:
            l1 = None
            if kind == KGE_TRANSE:
                if zt_bf16:   # norms of the rows as the kernels read them
                    l1 = zt.float().abs().sum(1).contiguous()
                else:
                    l1 = torch.empty(n, dtype=torch.float32, device=z.device)
                    L.call("bkg_row_l1norm_f32", L.ptr(z), z.stride(0), n, d, L.ptr(l1), L.stream())
            # the incidence structures: (edge_index, edge_type, g, rec, incidence, triplets, rows of dz, first table row)
            if owned is None:
                pos_ei, pos_et, neg_ei, neg_et = ctx.graphs
                both = ((pos_ei, pos_et, g_pos, rec_pos), (neg_ei, neg_et, g_neg, rec_neg))
                groups = [(ei, et, g, rec, (incidence_of(ei, n) if ei.size(1) else None), ei.size(1), n, 0)
                          for ei, et, g, rec in (both[1:] if ctx.onepass else both)]    # (one-pass: dz_pos holds the positives' part)
            else:
                groups = [(None, ts.own_type, None, rec, (ts.incidence if ts.n_own + ts.n_foreign else None),
                           ts.n_own + ts.n_foreign, ts.n_local, ts.row0)
                          for ts, rec in ((owned[0], rec_pos), (owned[1], rec_neg))]
            n_rows = groups[0][6]
            dz = torch.empty(n_rows, d, dtype=torch.float32, device=z.device)
            first = True
            base = dz_pos if ctx.onepass else None      # what the first launch over dz adds to (it stays intact)
            live = [gi for gi, gr in enumerate(groups)
                    if gr[5] and gr[4] is not None and (2 * gr[5] if owned is None else gr[4].n_inc)]
            # the last launch over dz applies the upstream gradient and the L2 term on its way out (the stream form's tail)
            tail = None
            if (KGE_BWD_TAIL and live and not ctx.padded and gloss.numel() == 1 and dz.stride(0) % 4 == 0
                    and (not ctx.reg or (z_reg is not None and z_reg.dtype == torch.float32 and z_reg.shape == dz.shape
                                         and z_reg.stride(1) == 1 and z_reg.stride(0) % 4 == 0 and z_reg.data_ptr() % 16 == 0))):
                ratio = 1.0 if ctx.reg_rows is None else dz.size(0) / ctx.reg_rows
                tail = (gloss.reshape(1).float().contiguous(), z_reg if ctx.reg else None,
                        2.0 * ctx.reg / d * ratio / max(dz.size(0), 1))
            tailed = False
            for gi, (ei, et, g, rec, inc, t, _, row0) in enumerate(groups):
                if t == 0 or inc is None:
                    continue
                n_inc = 2 * t if owned is None else inc.n_inc
                if n_inc == 0:
                    continue
                sp = inc.split.use() if inc.split is not None else None
                partials = sp.partials(d, z.device) if sp is not None else None
                # algorithmic bytes: per incidence the OTHER endpoint's row + the 16-byte record + the incidence id; per
                # node the fp32 gradient row (read back too when accumulating the second triplet set) + its row pointer
                by = n_inc * (d * zt.element_size() + 16 + 4) + n_rows * (4 * d * (1 if first else 2) + 4)
                stream = (KGE_BWD_STREAMS and kind == KGE_DISTMULT and sp is not None and rec is not None
                          and (rel_emb.size(0) + 1) * d * 4 <= 65536 and n < (1 << 31))
                if stream:
                    # the lists' streams in list order: other endpoint and relation id once per structure (cached on it:
                    # the positives keep theirs, fresh negatives build them with their g), g every step -- then the list
                    # reduce runs like the SpMM's (csrc/kge_bwd_nodes.hip, "stream form")
                    gs = torch.empty(n_inc, dtype=torch.float32, device=z.device)
                    # (priced like the one-pass form: the streams' extra 16 bytes per incidence are this form's own traffic)
                    with _Timed("kge_bwd_nodes" if not ctx.padded else None, by, d=d):
                        # the two fixed streams belong to (this incidence structure, this relation-id tensor OBJECT at this
                        # version): an id tensor re-allocated at the same address (fix_edge_id's full_like every step) is
                        # another object and rebuilds them
                        held = getattr(inc, "streams_of", None)
                        fresh = held is None or held[0]() is not et or held[1] != et._version
                        if fresh:
                            inc.other = torch.empty(n_inc, dtype=torch.int32, device=z.device)
                            inc.relid = torch.empty(n_inc, dtype=torch.int32, device=z.device)
                            inc.streams_of = (weakref.ref(et), et._version)
                        L.call("bkg_kge_incidence_streams", L.ptr(inc.perm), n_inc, t, None, None, None, L.ptr(rec),
                               None, L.ptr(inc.other) if fresh else None, L.ptr(inc.relid) if fresh else None, L.ptr(gs),
                               L.stream())
                        tl = tail if (tail is not None and gi == live[-1]) else None
                        src = base if first else None
                        L.call("bkg_kge_bwd_stream_f32", L.ptr(zt), zt_bf16, zt.stride(0), n_rows, d, L.ptr(rel_emb),
                               rel_emb.stride(0), rel_emb.size(0), L.ptr(inc.other), L.ptr(inc.relid), L.ptr(gs), L.ptr(dz),
                               dz.stride(0), L.sref(sp.c), L.ptr(partials), 0 if (first and src is None) else 1,
                               L.ptr(tl[0]) if tl else None, L.ptr(tl[1]) if tl else None,
                               tl[1].stride(0) if (tl and tl[1] is not None) else 0, tl[2] if tl else 0.0,
                               L.ptr(src), src.stride(0) if src is not None else 0, L.stream())
                        tailed = tl is not None
                        base = None
                    first = False
                    continue
                if base is not None:          # (the one-pass form's positives + a group on the older path: start from a copy)
                    dz.copy_(base)
                    base, first = None, False
                with _Timed("kge_bwd_nodes" if not ctx.padded else None, by, d=d):
                    L.call("bkg_kge_bwd_nodes_f32", kind, L.ptr(zt), zt_bf16, zt.stride(0), n_rows, d,
                           L.ptr(ei[0]) if ei is not None else None, L.ptr(ei[1]) if ei is not None else None,
                           L.ptr(et) if ei is not None else None, t, L.ptr(rel_emb), rel_emb.stride(0), rel_emb.size(0),
                           L.ptr(g) if g is not None else None, L.ptr(rec), L.ptr(l1), L.ptr(inc.rowptr),
                           L.ptr(inc.perm), L.ptr(dz), dz.stride(0), L.sref(sp.c if sp is not None else None),
                           L.ptr(partials), 0 if first else 1, row0, L.stream())
                first = False
            if base is not None:              # no other group ran: dz is the positives' part
                dz.copy_(base)
            elif first:
                dz.zero_()
            if tailed:
                pass             # (scaled, L2 term added: the last stream launch's tail)
            elif ctx.padded:     # dz * gloss + the L2 term's 2 reg z / (real rows * d) on the real rows: one launch
                gl = gloss.reshape(1).contiguous()
                L.call("bkg_scale_add_rows_f32", L.ptr(dz), dz.stride(0), L.ptr(z_reg), z_reg.stride(0), n, d, L.ptr(gl),
                       2.0 * ctx.reg / d, L.ptr(ctx.n_real), L.stream())
            elif ctx.reg and gloss.numel() == 1 and z_reg.dtype == torch.float32 and z_reg.shape == dz.shape:
                # dz * gloss + (2 reg / numel) * gloss * z in one in-place launch (dz is this op's own buffer); the kernel
                # divides by the rows it is given: a partitioned step's mean runs over reg_rows rows instead
                ratio = 1.0 if ctx.reg_rows is None else dz.size(0) / ctx.reg_rows
                if not dz.is_contiguous() or dz.data_ptr() % 16:
                    dz = dz.contiguous()
                gl = gloss.reshape(1).float().contiguous()
                L.call("bkg_scale_add_rows_f32", L.ptr(dz), dz.stride(0), L.ptr(z_reg), z_reg.stride(0), dz.size(0), d,
                       L.ptr(gl), 2.0 * ctx.reg / d * ratio, None, L.stream())
            else:
                dz = dz * gloss
                if ctx.reg:
                    z_rows = z_reg.size(0) if ctx.reg_rows is None else ctx.reg_rows
                    dz = torch.addcmul(dz, z_reg, gloss * (2.0 * ctx.reg / (z_rows * z_reg.size(1))))
        drel_out = None
        if ctx.needs_input_grad[1]:
            if ctx.padded:
                drel_out = drel.clone()
                gl = gloss.reshape(1).contiguous()
                L.call("bkg_scale_add_rows_f32", L.ptr(drel_out), drel_out.stride(0), L.ptr(rel_emb), rel_emb.stride(0),
                       rel_emb.size(0), d, L.ptr(gl), 2.0 * ctx.reg / d, None, L.stream())
            elif KGE_LOSS_FINISH and gloss.numel() == 1 and drel.is_contiguous():
                drel_out = torch.empty_like(drel)            # drel * gloss + (2 reg / numel) * gloss * rel_emb: one launch
                gl = gloss.reshape(1).float().contiguous()
                L.call("bkg_scale_add_rows_out_f32", L.ptr(drel), drel.stride(0), L.ptr(rel_emb), rel_emb.stride(0),
                       rel_emb.size(0), d, L.ptr(gl), 2.0 * ctx.reg / d, L.ptr(drel_out), drel_out.stride(0), L.stream())
            else:
                drel_out = drel * gloss
                if ctx.reg:
                    drel_out = torch.addcmul(drel_out, rel_emb, gloss * (2.0 * ctx.reg / rel_emb.numel()))
        return dz, drel_out, None, None, None, None, None, None, None, None, None, None, None, None


def kge_bce_loss_owned(z_local, rel_emb, pos: OwnedTriplets, neg: OwnedTriplets, kind, comm, total, reg=0.0, reg_rows=None):
    """The same op over a node-partitioned rank's triplets (``_KgeBceLoss`` with ``owned``): ``z_local`` is the rank's
    shard, the triplets are table slots, ``total`` the count the BCE sum is divided by (global count / world)."""
    L.require_cuda(z_local, rel_emb)
    if reg and reg_rows is None:
        raise ValueError("the L2 term of a partitioned step needs reg_rows = N / world (the rows its mean runs over)")
    if comm is not None and not comm.live:
        comm = None
    e2 = torch.empty(2, 0, dtype=torch.int64, device=z_local.device)
    return _KgeBceLoss.apply(_f32c(z_local), _f32c(rel_emb), e2, e2[0], e2, e2[0], kind, comm, total, float(reg), None,
                             None, reg_rows, (pos, neg))


def kge_bce_loss(z, rel_emb, pos_edge_index, pos_edge_type, neg_edge_index, neg_edge_type, kind, total=None,
                 reg=0.0, inv_total=None, rows=None, reg_rows=None, lazy_scores=False):
    """(Node-partitioned training: ``kge_bce_loss_owned``.)  ``total``: the count the BCE sum is divided by (default: this call's triplets);
    ``reg``: adds ``reg * (mean(z^2) + mean(rel_emb^2))`` (the reference's L2 term) inside the same op.
    ``inv_total`` / ``rows`` (padded mini-batches of a captured step, biomedkg_amd/sampled.py): the BCE sum is multiplied
    by the DEVICE scalar ``inv_total``, triplets with relation id ``R`` (one past the real ones) are pads, and the mean
    of ``z^2`` runs over the first ``rows`` rows (an int64 DEVICE scalar).  ``reg_rows`` (partitioned steps): the number
    of rows the mean of ``z^2`` is taken over when ``z`` is a shard (N / world: the bucket average is the whole mean).
    ``lazy_scores``: a fourth value comes back -- None, or the inverse permutation of the sorted frame the positives' scores
    were left in (``scores_in_order`` puts them in the caller's order: a training step only does when somebody asks)."""
    L.require_cuda(z, rel_emb, pos_edge_index, pos_edge_type, neg_edge_index, neg_edge_type, inv_total)
    for ei, et, what in ((pos_edge_index, pos_edge_type, "positive"), (neg_edge_index, neg_edge_type, "negative")):
        if ei.dim() != 2 or ei.size(0) != 2 or et.dim() != 1 or et.numel() != ei.size(1):
            # (the reference fails here too -- a broadcast error in decode -- e.g. when negative_sampling comes back
            # with fewer pairs than neg_ratio * E on a tiny dense batch; the kernels index one by the other)
            raise ValueError(f"{what} triplets: edge_index {tuple(ei.shape)} and edge_type {tuple(et.shape)} do not match")
    if inv_total is not None and (inv_total.dtype != torch.float32 or rows is None or rows.dtype != torch.int64):
        raise ValueError("inv_total (padded mini-batches): a float32 device scalar together with rows = the real row "
                         "count as an int64 device scalar")
    _FRAME["lazy"] = bool(lazy_scores)
    try:
        out = _KgeBceLoss.apply(_f32c(z), _f32c(rel_emb), pos_edge_index.contiguous(), pos_edge_type.contiguous(),
                                neg_edge_index.contiguous(), neg_edge_type.contiguous(), kind, None, total, float(reg),
                                inv_total, rows, reg_rows)
    finally:
        _FRAME["lazy"] = False
    if lazy_scores:        # (loss, positives' scores possibly in frame order, negatives' scores, the inverse permutation or None)
        inverse, _FRAME["inverse"] = _FRAME["inverse"], None
        lazy_neg, _FRAME["lazy_neg"] = _FRAME.get("lazy_neg"), None
        if lazy_neg is not None:       # the negatives' one-pass walk wrote no scores: a LazyScores in their place
            out = (out[0], out[1], lazy_neg)
        return out + (inverse,)
    return out


# ----------------------------------------------------------------------------- NT-Xent
class _NtXent(torch.autograd.Function):
    """GRACE loss: DualBranchContrast(InfoNCE(tau), L2L, intraview_negs=True)(h1, h2) -- reference
    gcl_module.py:171-173,186-190.  Three launches forward (normalise, Gram tiles, fixed-order finish with the loss sum), two
    backward (csrc/ntxent2.hip); ``operands``: "fp32" = exact fp32 MFMA, "bf16" = bf16 MFMA operands with fp32
    exp / sums (the matmul precision mode picks it)."""

    @staticmethod
    def forward(ctx, h1, h2, tau, bf16):
        n, h = h1.shape
        m, dev = 2 * n, h1.device
        hpad = L.lib().bkg_ntxent_hpad(h, bf16)
        if hpad < 0:
            raise ValueError(f"ntxent_loss: embedding width {h} > 256 is not supported")
        zn = torch.empty(m, hpad, dtype=torch.float32, device=dev)
        zn16 = torch.empty(m, hpad, dtype=torch.bfloat16, device=dev) if bf16 else None
        norm = torch.empty(m, dtype=torch.float32, device=dev)
        posdot = torch.empty(n, dtype=torch.float32, device=dev)
        L.call("bkg_ntxent_prepare", L.ptr(h1), h1.stride(0), L.ptr(h2), h2.stride(0), n, h, hpad, tau, L.ptr(zn), L.ptr(zn16),
               L.ptr(norm), L.ptr(posdot), L.stream())
        ws = torch.empty(L.lib().bkg_ntxent_forward_workspace_floats(m, h, bf16), dtype=torch.float32, device=dev)
        den = torch.empty(2, m, dtype=torch.float32, device=dev)          # D and 1 / D
        loss = torch.empty(1, dtype=torch.float32, device=dev)             # sum(term) / m, added inside the finish launch
        L.call("bkg_ntxent_forward", L.ptr(zn16 if bf16 else zn), bf16, m, h, tau, L.ptr(posdot), L.ptr(ws), L.ptr(den[0]),
               L.ptr(den[1]), None, L.ptr(loss), L.stream())
        ctx.tau, ctx.n, ctx.h, ctx.bf16 = tau, n, h, bf16
        ctx.save_for_backward(zn, zn16, norm, den)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        zn, zn16, norm, den = ctx.saved_tensors
        n, h, tau, bf16 = ctx.n, ctx.h, ctx.tau, ctx.bf16
        m, dev = 2 * n, zn.device
        g = g.reshape(1).to(torch.float32).contiguous()
        ws = torch.empty(L.lib().bkg_ntxent_backward_workspace_floats(m, h, bf16), dtype=torch.float32, device=dev)
        dh = torch.empty(2, n, h, dtype=torch.float32, device=dev)
        L.call("bkg_ntxent_backward", L.ptr(zn16 if bf16 else zn), bf16, L.ptr(zn), L.ptr(norm), L.ptr(den[1]), L.ptr(g), m, h,
               tau, L.ptr(ws), L.ptr(dh[0]), h, L.ptr(dh[1]), h, L.stream())
        return dh[0], dh[1], None, None


class NtXentKernels:
    """The NT-Xent arithmetic behind ``partition.ntxent_partitioned`` (csrc/ntxent2.hip, row-partitioned entry points);
    the gloo tests inject a dense torch restatement with the same four methods."""

    def __init__(self, operands=None):
        self.bf16 = int((operands or ("bf16" if _PRECISION == 1 else "fp32")) == "bf16")

    def normalize(self, h1, h2):
        """-> ``zn [2n, hpad]`` (fp32, view 1 rows then view 2 rows), ``norm [2n]``, positive dots ``[n]``"""
        h1, h2 = _f32c(h1), _f32c(h2)
        n, h = h1.shape
        self.h = h
        hpad = L.lib().bkg_ntxent_hpad(h, self.bf16)
        if hpad < 0:
            raise ValueError(f"ntxent: embedding width {h} > 256 is not supported")
        zn = torch.empty(2 * n, hpad, dtype=torch.float32, device=h1.device)
        norm = torch.empty(2 * n, dtype=torch.float32, device=h1.device)
        posdot = torch.empty(n, dtype=torch.float32, device=h1.device)
        L.call("bkg_ntxent_prepare", L.ptr(h1), h1.stride(0), L.ptr(h2), h2.stride(0), n, h, hpad, 1.0, L.ptr(zn), None,
               L.ptr(norm), L.ptr(posdot), L.stream())
        return zn, norm, posdot

    def _operand(self, z_all, tau):
        # the bf16 image carries sqrt(log2(e) / tau) on every row (include/biomedkg_hip.h: bkg_ntxent_bf16_scale)
        return to_bf16(z_all * float(L.lib().bkg_ntxent_bf16_scale(tau))) if self.bf16 else z_all

    def row_sums(self, z_all, anchors, tau):
        m, na = z_all.size(0), anchors.numel()
        ws = torch.empty(L.lib().bkg_ntxent_rows_forward_workspace_floats(m, na, self.h, self.bf16), dtype=torch.float32,
                         device=z_all.device)
        den = torch.empty(na, dtype=torch.float32, device=z_all.device)
        z16 = self._operand(z_all, tau)
        # kept for the backward of THIS forward only: keyed on the saved tensor itself (one kernels object may serve
        # several forwards before their backwards -- two GRACE losses in a step, gradient accumulation)
        stash = self.__dict__.setdefault("_z16", {})
        if len(stash) >= 4:
            stash.pop(next(iter(stash)))
        stash[(z_all.data_ptr(), z_all._version, float(tau))] = z16
        L.call("bkg_ntxent_rows_forward", L.ptr(z16), self.bf16, m, self.h, tau, L.ptr(anchors), na, L.ptr(ws),
               L.ptr(den), L.stream())
        return den

    def row_grads(self, z_all, anchors, den_all, norm, tau, scale, n_half):
        """-> ``dh [n_anchor, h]``: d(loss)/d(un-normalised anchor rows) * scale (``scale``: device scalar tensor)"""
        m, na = z_all.size(0), anchors.numel()
        ws = torch.empty(L.lib().bkg_ntxent_rows_backward_workspace_floats(m, na, self.h, self.bf16), dtype=torch.float32,
                         device=z_all.device)
        rden = (1.0 / den_all).contiguous()
        dh = torch.empty(na, self.h, dtype=torch.float32, device=z_all.device)
        z16 = self.__dict__.get("_z16", {}).pop((z_all.data_ptr(), z_all._version, float(tau)), None)
        if z16 is None or z16.size(0) != m:
            z16 = self._operand(z_all, tau)
        scale = scale.reshape(1).to(torch.float32).contiguous()
        L.call("bkg_ntxent_rows_backward", L.ptr(z16), self.bf16, L.ptr(z_all), L.ptr(norm), L.ptr(rden), L.ptr(scale), m,
               self.h, tau, L.ptr(anchors), na, n_half, 1.0, L.ptr(ws), L.ptr(dh[:n_half]), self.h,
               L.ptr(dh[n_half:]), self.h, L.stream())
        return dh


def ntxent_loss(h1, h2, tau=0.2, operands=None):
    """``operands``: None = follow the matmul precision mode, or "fp32" / "bf16" explicitly."""
    L.require_cuda(h1, h2)
    if operands is None:
        operands = "bf16" if _PRECISION == 1 else "fp32"
    if h1.shape != h2.shape or h1.dim() != 2:
        raise ValueError("ntxent_loss expects two [N, h] views")
    if h1.size(1) > 256:
        raise ValueError(f"ntxent_loss: embedding width {h1.size(1)} > 256 is not supported")
    h1, h2 = _f32c(h1), _f32c(h2)
    if h1.size(0) == 0:
        return (h1.sum() + h2.sum()) * 0.0
    return _NtXent.apply(h1, h2, float(tau), int(operands == "bf16"))


# ----------------------------------------------------------------------------- modality fusion
FUSION_CHUNK_NODES = int(__import__("os").environ.get("BKG_FUSION_CHUNK", "8192"))   # 2 * 8192 rows x 2304 fp32 = 151 MB of
#                            q|k|v per chunk: stays in the 256 MiB Infinity Cache (4,096: 3.4 ms first call, 8,192: 3.1)
REDAF_CHUNK_NODES = int(__import__("os").environ.get("BKG_REDAF_CHUNK", "16384"))    # 2 * 16384 rows x 768 fp32 = 100 MB of t
#                            (4,096-node chunks: 1.47 ms per call at N = 130k -- products too small to fill the chip --, 16,384: 1.0)


class _GgdBce(torch.autograd.Function):
    """``binary_cross_entropy_with_logits(cat(pos, neg), cat(ones, zeros))`` (mean) for GGD's two score vectors, one
    forward and one backward launch (csrc/gcl_loss.hip); ``n_dev``: the masked form of padded mini-batches."""

    @staticmethod
    def forward(ctx, pos, neg, n_dev):
        L.require_cuda(pos, neg, n_dev)
        n = pos.numel()
        partial = torch.empty(int(L.lib().bkg_ggd_bce_partials()), dtype=torch.float32, device=pos.device)
        out = torch.empty((), dtype=torch.float32, device=pos.device)
        L.call("bkg_ggd_bce_f32", L.ptr(pos), pos.stride(0) if n else 1, L.ptr(neg), neg.stride(0) if n else 1, n,
               L.ptr(n_dev), L.ptr(partial), L.ptr(out), L.stream())
        ctx.save_for_backward(pos, neg, n_dev)
        return out

    @staticmethod
    def backward(ctx, gout):
        pos, neg, n_dev = ctx.saved_tensors
        n = pos.numel()
        d = torch.empty(n, 2, dtype=torch.float32, device=pos.device)       # [d pos | d neg]: the layout of the score table
        g = _f32c(gout).reshape(1)
        L.call("bkg_ggd_bce_bwd_f32", L.ptr(pos), pos.stride(0) if n else 1, L.ptr(neg), neg.stride(0) if n else 1, n,
               L.ptr(n_dev), L.ptr(g), L.ptr(d), 2, L.ptr(d[:, 1]) if n else L.ptr(d), 2, L.stream())
        return d[:, 0], d[:, 1], None


def ggd_bce_ok(pos, neg):
    return (pos.is_cuda and neg.is_cuda and pos.dim() == 1 and neg.shape == pos.shape and pos.dtype == torch.float32
            and neg.dtype == torch.float32 and pos.numel() > 0)


def ggd_bce(pos, neg, n_dev=None):
    """GGD's loss over the positive / corrupted score vectors (any element stride); ``n_dev``: int64 device scalar, rows
    below it count (padded mini-batches)."""
    return _GgdBce.apply(pos, neg, n_dev)


REDAF_ONE_LAUNCH = __import__("os").environ.get("BKG_REDAF_ONE_LAUNCH", "1") != "0"


def _fusion_rows(x, n_out):
    """``x [n, M, D]`` as the ``[n M, D]`` operand of the fuser's product with an ``[n_out, D]`` weight: fp32, or -- bf16
    mode, constant input features, a shape the streaming product takes -- a bf16 copy cached on the tensor (that product
    re-reads its A operand once per 128-column tile of the weight: 6 (ReDAF) or 18 (q|k|v) times 400 MB instead of 800)."""
    n, m, d = x.shape
    if x.dtype == torch.bfloat16:       # rows that already are a bf16 table (bf16 mode takes them as they are)
        if bf16_storage() and x.is_contiguous() and _bf16_operand_ok(max(n * m, _TALL_ROWS), n_out, d):
            return x.view(n * m, d)
        x = x.float()
    x2 = _f32c(x).view(n * m, d)
    chunk_rows = min(n, FUSION_CHUNK_NODES, REDAF_CHUNK_NODES) * m          # every chunk of it must be a tall operand too
    if (bf16_storage() and not x.requires_grad and n * m >= 4096 and x.is_contiguous()
            and _bf16_operand_ok(chunk_rows, n_out, d)):
        from .graph import _cached
        return _cached(x, "bf16_rows", (), lambda: to_bf16(x2))
    return x2


PROJECTED_ATTENTION = __import__("os").environ.get("BKG_PROJECTED_ATTENTION", "1") != "0"


def _exact_gemm(a, b, trans_a=False, trans_b=False):
    """A small fp32 product at exact precision whatever the mode (constants of a frozen module)."""
    a, b = _f32c(a), _f32c(b)
    m, k = (a.size(1), a.size(0)) if trans_a else (a.size(0), a.size(1))
    n = b.size(0) if trans_b else b.size(1)
    out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    L.call("bkg_gemm_f32", int(trans_a), int(trans_b), m, n, k, 1.0, L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), 0.0,
           L.ptr(out), out.stride(0), None, ACT_NONE, None, 0, 1.0, 0, L.stream())
    return out


def _attention_fusion_projected(x2, w_qkv, b_qkv, n, d):
    """bf16 mode, two modalities: AttentionFusion (reference fusion.py:17-31) through two identities of its own arithmetic,
    nothing skipped or approximated beyond the mode's bf16 operands:

    * ``q_a . k_b = x_a^T (Wq^T Wk) x_b + (Wq^T bk) . x_a + (Wk^T bq) . x_b + bq . bk`` -- the fuser is frozen (SURVEY
      0.3-1), so ``Wq^T Wk`` is a constant of the run: ONE ``[2n, D] x [D, D]`` product ``T = X (Wq^T Wk)`` replaces the
      q and the k projection, and the four scores of a node are row dots of T with X;
    * ``sum_b c_b (Wv x_b + bv) = Wv (sum_b c_b x_b) + bv`` because the averaged attention weights ``c_b`` sum to one: the
      V projection runs over the ``n`` COMBINED rows instead of the ``2n`` token rows.

    1.5 D^2 multiply-adds per token instead of 3 D^2, two streaming passes over X, no ``[2n, 3D]`` round trip:
    product (csrc/gemm_xw.hip) -> scores / softmax / combine (csrc/fusion.hip ``attn_scores_combine_kernel``) -> product.
    The exact-fp32 mode keeps the reference's literal form (three projections, then the 2 x 2 attention)."""
    wq, wk, wv = w_qkv[:d], w_qkv[d:2 * d], w_qkv[2 * d:]
    bq, bk, bv = b_qkv[:d], b_qkv[d:2 * d], b_qkv[2 * d:]

    def constants():
        # constants of the frozen fuser: three small exact products and one host read, once per weight version (they were
        # recomputed -- and the host read stalled the launch queue -- on every call with new features: ~0.15 ms of 1.3)
        return (_exact_gemm(wq, wk, trans_a=True),                        # Wq^T Wk  [D, D]
                _exact_gemm(bq.view(1, d), wk).view(d),                   # Wk^T bq
                _exact_gemm(bk.view(1, d), wq).view(d),                   # Wq^T bk
                float((bq.double() * bk.double()).sum()),
                wv)                                                       # (the SAME view object every call: its bf16 image stays cached on it)
    from .graph import _cached
    mqk, u, w, c0, wv = _cached(w_qkv, "attn_projected", (b_qkv.data_ptr(), b_qkv._version, d), constants)
    t = gemm(x2, mqk, out_dtype=torch.bfloat16)                           # [2n, D] bf16
    xc = torch.empty(n, d, dtype=torch.bfloat16, device=x2.device)
    L.call("bkg_attn_scores_combine_bf16", L.ptr(t), t.stride(0), L.ptr(x2), x2.stride(0), n, d, L.ptr(u), L.ptr(w), c0,
           L.ptr(xc), xc.stride(0), L.stream())
    if _xw_ok(xc, n, d, d):
        return gemm(xc, wv, trans_b=True, bias=bv)                        # [n, D] fp32
    return gemm(xc.float(), wv, trans_b=True, bias=bv)


def attention_fusion(x, w_qkv, b_qkv):
    """AttentionFusion.forward (reference fusion.py:17-31): one QKV GEMM + fused MxM attention/mean.
    ``x [n, M, D]``, ``w_qkv [3D, D]``, ``b_qkv [3D]`` -> ``[n, D]``.  Forward only (the reference never
    trains the fuser: its parameters are outside every optimizer).  Runs in node chunks so that the
    ``[M n, 3D]`` projection (2.4 GB at n = 130k) never makes the HBM round trip between its two kernels: each chunk's
    q|k|v is consumed out of the Infinity Cache by the attention / mean kernel right behind the product."""
    L.require_cuda(x, w_qkv, b_qkv)
    n, m, d = x.shape
    x2 = _fusion_rows(x, w_qkv.size(0))
    if PROJECTED_ATTENTION and x2.dtype == torch.bfloat16 and m == 2 and d % 8 == 0 and d <= 2048 and _xw_ok(x2, 2 * n, d, d):
        return _attention_fusion_projected(x2, w_qkv, b_qkv, n, d)
    out = torch.empty(n, d, dtype=torch.float32, device=x.device)
    step = FUSION_CHUNK_NODES if n > 2 * FUSION_CHUNK_NODES else max(n, 1)
    qkv = None
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        rows = (hi - lo) * m
        buf = None if qkv is None or qkv.size(0) < rows else qkv[:rows]
        qkv_c = gemm(x2[lo * m: hi * m], w_qkv, trans_b=True, bias=b_qkv, out=buf)
        if qkv is None:
            qkv = qkv_c
        L.call("bkg_attn_fuse_f32", L.ptr(qkv_c), qkv_c.stride(0), hi - lo, m, d, L.ptr(out[lo:hi]), out.stride(0),
               L.stream())
    return out


def redaf_fusion(x, w_t, b_t, scale, drop: Dropout = None):
    """ReDAF.forward (reference fusion.py:49-90): relu(Linear_T(x)) in the GEMM epilogue, then
    mean_m relu(dropout(t * scale[m])) with ``scale = modal_weights * zeta`` ``[M, D]``.  Dropout makes the result new
    every training step (nothing to cache, unlike AttentionFusion), so the ``[M n, D]`` transform must not make an HBM
    round trip per step: like ``attention_fusion`` it runs in node chunks -- 4,096 nodes = 8,192 x 768 fp32 = 25 MB --
    and the combine kernel reads each chunk's ``t`` out of the Infinity Cache right behind the product (at n = 130,000:
    one 25 MB scratch instead of an 800 MB intermediate; the keep hash is indexed by the GLOBAL row, so the masks
    do not depend on the chunking)."""
    L.require_cuda(x, w_t, b_t, scale)
    n, m, d = x.shape
    x2 = _fusion_rows(x, w_t.size(0))
    scale = _f32c(scale)
    out = torch.empty(n, d, dtype=torch.float32, device=x.device)
    keep, p, seed, sid = None, 0.0, 0, 0
    if drop is not None and drop.p > 0:
        p, seed, sid = drop.p, drop.seed, drop.stream_id      # stream_id: the provider's per-call counter (fresh mask per call)
        if drop.keep is not None:
            keep = drop.keep.to(torch.uint8).contiguous()
    if m == 2 and REDAF_ONE_LAUNCH and n > 0 and _xw_ok(x2, n * m, w_t.size(0), d) and w_t.size(0) == d:
        # two modalities, bf16 mode, a shape the streaming product takes: the whole fuser is the product's epilogue
        # (bkg_redaf_fused_xw) -- no [2n, D] transform anywhere, one launch per call; bit-equal with the chunked form below
        wt16 = _weight_bf16(_f32c(w_t), False)
        L.call("bkg_redaf_fused_xw", n * m, d, d, L.ptr(x2), int(x2.dtype == torch.bfloat16), x2.stride(0), L.ptr(wt16),
               wt16.stride(0), L.ptr(_f32c(b_t)), L.ptr(scale), L.ptr(keep), p, seed, sid, L.ptr(out), out.stride(0), L.stream())
        return out
    step = REDAF_CHUNK_NODES if n > 2 * REDAF_CHUNK_NODES else max(n, 1)
    t = None
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        rows = (hi - lo) * m
        buf = None if t is None or t.size(0) < rows else t[:rows]
        t_c = gemm(x2[lo * m: hi * m], w_t, trans_b=True, bias=b_t, act=ACT_RELU, out=buf)
        if t is None:
            t = t_c
        L.call("bkg_redaf_combine_f32", L.ptr(t_c), t_c.stride(0), hi - lo, m, d, L.ptr(scale),
               L.ptr(keep[lo * m: hi * m]) if keep is not None else None, p, seed, sid, lo * m, L.ptr(out[lo:hi]),
               out.stride(0), L.stream())
    return out
