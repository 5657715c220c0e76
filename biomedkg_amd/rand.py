"""Randomness providers for the stochastic steps of the hot path.

The reference draws every mask / permutation with torch (on whatever device ``x`` lives on) and the
negative samples with python ``random`` (reference ``biomedkg/model/gcl.py:40-43,75-76``,
``biomedkg/kge_module.py:86-102``, PyG ``negative_sampling``).  RNG streams cannot be matched
across devices, so every draw goes through a provider:

* ``DeviceRand`` (default): draws on the GPU -- torch's device generator for feature / edge masks
  and permutations, an in-kernel Philox stream for layer dropout (no mask ever touches HBM), and
  the device rejection sampler for negatives.
* ``ReplayRand``: replays a tape of draws recorded by a CPU run (tape format:
  ``[(kind, tensor), ...]`` with kinds feat_keep / edge_keep / drop_keep / perm / coin / neg_edges)
  -- this is how the parity tests feed identical randomness to the HIP path and to the checker.
"""
from __future__ import annotations

import torch

from .ops import Dropout


_PERM_SIZES = {}     # (n, device) -> the device scalar bkg_randperm_dev reads its extent from


def max_below(ids, bound):
    """Are all ids in [0, bound)?  (One host read, once per id tensor: the caller keeps the answer on the tensor.)"""
    return ids.numel() == 0 or (int(ids.max()) < bound and int(ids.min()) >= 0)


class DeviceRand:
    device_draws = True     # draws are made on the device in whatever order / shape the fast path needs (no tape)

    def __init__(self, seed=None):
        self.seed = int(torch.initial_seed() if seed is None else seed) & ((1 << 63) - 1)
        self.counter = 0

    def feat_keep(self, x, p):
        return torch.rand_like(x) >= p

    def masked(self, x, p):
        """``x * feat_keep(x, p)``; in bf16 mode a bf16 table straight from one pass over ``x`` (in-kernel Philox)."""
        from . import ops
        if ops.features_bf16_ok(x) and x.is_cuda:
            self.counter += 1
            return ops.masked_bf16(x, p, self.seed, self.counter & 0xFFFFFFFF)
        return x * self.feat_keep(x, p)

    def edge_keep(self, num_edges, p, device):
        return torch.rand(num_edges, device=device) >= p

    def dropout(self, shape, p, device) -> Dropout:
        self.counter += 1
        return Dropout(p, keep=None, seed=self.seed, stream_id=self.counter & 0xFFFFFFFF)

    def perm(self, n, device):
        """``torch.randperm(n)`` of the corruption steps (reference gcl.py:22-24,70-71) as a keyed bijection of the index
        range -- ONE launch (csrc/negsample.hip) instead of torch's random keys + radix sort (ten launches at n = 130 k);
        under a captured step the library's replay counter re-keys it."""
        device = torch.device(device)
        if device.type != "cuda" or n == 0:
            return torch.randperm(n, device=device)
        from . import _lib as L
        key = (n, str(device))
        n_dev = _PERM_SIZES.get(key)
        if n_dev is None:
            n_dev = _PERM_SIZES[key] = torch.tensor([n], dtype=torch.int64, device=device)
        out = torch.empty(n, dtype=torch.int64, device=device)
        L.call("bkg_randperm_dev", L.ptr(n_dev), n, self._draw_seed(), None, L.ptr(out), L.stream())
        return out

    def coin(self) -> float:
        return float(torch.rand(1).item())

    def _draw_seed(self):
        self.counter += 1
        return (self.seed * 0x9E3779B97F4A7C15 + self.counter * 0xD1B54A32D192ED03) & ((1 << 63) - 1)

    def shuffle(self, src):
        """``src[randperm(len(src))]`` for an int64 vector through a keyed bijection of the index range
        (csrc/negsample.hip: no sort behind it)."""
        from . import _lib as L
        src = src.contiguous()
        if src.dtype != torch.int64 or src.dim() != 1:
            return src[torch.randperm(src.size(0), device=src.device)]
        out = torch.empty_like(src)
        L.call("bkg_permute_gather_i64", L.ptr(src), src.numel(), self._draw_seed(), L.ptr(out), L.stream())
        return out

    def shuffle_repeat(self, src, repeat):
        """``src.repeat(repeat)[randperm(repeat * len(src))]`` (the negatives' relation ids, reference kge_module.py:97-100)
        in one launch, without the repeated copy; takes the draw ``shuffle`` would have taken."""
        from . import _lib as L
        src = src.contiguous()
        if src.dtype != torch.int64 or src.dim() != 1 or not src.is_cuda or repeat < 1:
            return self.shuffle(src.repeat(repeat))
        out = torch.empty(src.numel() * int(repeat), dtype=torch.int64, device=src.device)
        small = getattr(src, "_bkg_u8", None)            # a one-byte image of ids < 256, kept on the tensor at this version
        if small is None or small[0] != src._version:
            small = None
            if src.numel() >= (1 << 16) and not torch.cuda.is_current_stream_capturing() and max_below(src, 256):
                img = torch.empty(src.numel(), dtype=torch.uint8, device=src.device)
                L.call("bkg_narrow_i64_u8", L.ptr(src), src.numel(), L.ptr(img), L.stream())
                small = src._bkg_u8 = (src._version, img)
            elif not torch.cuda.is_current_stream_capturing():
                src._bkg_u8 = (src._version, None)
        seed = self._draw_seed()
        if small is not None and small[1] is not None:
            L.call("bkg_permute_gather_repeat_u8", L.ptr(small[1]), src.numel(), int(repeat), seed, L.ptr(out), L.stream())
        else:
            L.call("bkg_permute_gather_repeat_i64", L.ptr(src), src.numel(), int(repeat), seed, L.ptr(out), L.stream())
        return out

    def grouped_types(self, edge_type, repeat, num_relations):
        """The relation ids of ``repeat`` negatives per positive, grouped by relation (``ops.grouped_types``: a uniformly
        random pairing with this provider's negatives, whose order is exchangeable); None where that form does not
        apply.  Takes the draw ``shuffle`` would have taken, so the streams behind it do not move."""
        from . import ops
        self._draw_seed()
        return ops.grouped_types(edge_type, repeat, num_relations)

    def neg_edges(self, edge_index, num_neg_samples):
        from .sampling import negative_sampling
        return negative_sampling(edge_index, num_neg_samples=num_neg_samples, seed=self._draw_seed())

    def skip(self, count):
        pass


class ReplayRand:
    def __init__(self, tape, device="cuda"):
        self.tape, self.pos, self.device = list(tape), 0, device

    def _next(self, kind):
        k, t = self.tape[self.pos]
        if k != kind:
            raise RuntimeError(f"tape mismatch at {self.pos}: have {k!r}, asked {kind!r}")
        self.pos += 1
        return t.to(self.device)

    def feat_keep(self, x, p):
        return self._next("feat_keep")

    def masked(self, x, p):
        return x * self.feat_keep(x, p)

    def edge_keep(self, num_edges, p, device):
        return self._next("edge_keep")

    def dropout(self, shape, p, device) -> Dropout:
        keep = self._next("drop_keep")
        return Dropout(p, keep=keep.reshape(shape))

    def perm(self, n, device):
        return self._next("perm")

    def shuffle(self, src):
        return src[self.perm(src.size(0), src.device)]

    def coin(self) -> float:
        return float(self._next("coin").item())

    def neg_edges(self, edge_index, num_neg_samples):
        return self._next("neg_edges")

    def skip(self, count):
        """Discard draws the HIP path does not need (e.g. GRACE's unused third encoder pass)."""
        self.pos += count

    def exhausted(self):
        return self.pos == len(self.tape)


_current = None


def current():
    global _current
    if _current is None:
        _current = DeviceRand()
    return _current


def reseed_default(seed):
    """Re-key the process-wide default provider (the Trainer stand-in gives every rank its own stream)."""
    global _current
    _current = DeviceRand(seed)


class use:
    def __init__(self, provider):
        self.provider = provider

    def __enter__(self):
        global _current
        self.prev, _current = _current, self.provider
        return self.provider

    def __exit__(self, *exc):
        global _current
        _current = self.prev
        return False
