"""Explicit-randomness providers for the CPU oracle (test infrastructure).

Every stochastic step of the hot path (feature masks, edge dropout, layer dropout, corruption
permutations, the GGD coin flip, negative samples, the negative-type permutation) draws through a
provider object so that the draws made by a CPU run can be recorded on a *tape* and replayed into
the HIP path: RNG streams cannot be matched across devices, the draws can.

Tape format (plain data, shared with ``biomedkg_amd.rand.ReplayRand``): a list of
``(kind, tensor)`` in call order, kinds:

* ``"feat_keep"``  bool ``[N, D]``  -- ``mask_feature``  keep mask (``rand_like(x) >= p``)
* ``"edge_keep"``  bool ``[E]``     -- ``dropout_edge``  keep mask (``rand(E) >= p``)
* ``"drop_keep"``  bool ``x.shape`` -- ``F.dropout``     keep mask (kept w.p. ``1-p``, scaled ``1/(1-p)``)
* ``"perm"``       int64 ``[n]``    -- ``torch.randperm(n)``
* ``"coin"``       float ``[1]``    -- ``torch.rand(1)``
* ``"neg_edges"``  int64 ``[2, K]`` -- result of ``negative_sampling``
"""
from __future__ import annotations

import random
from typing import List, Tuple

import torch

Tape = List[Tuple[str, torch.Tensor]]

# Captured at import so that golden generation may monkey-patch the torch entry points the
# reference calls directly (torch.rand / torch.randperm / F.dropout) to route through a provider.
_t_rand, _t_rand_like, _t_randperm = torch.rand, torch.rand_like, torch.randperm


class TorchRand:
    """Draws from the torch CPU global generator / python ``random`` (what the reference does)."""

    def feat_keep(self, x: torch.Tensor, p: float) -> torch.Tensor:
        return _t_rand_like(x) >= p

    def edge_keep(self, num_edges: int, p: float) -> torch.Tensor:
        return _t_rand(num_edges) >= p

    def drop_keep(self, x: torch.Tensor, p: float) -> torch.Tensor:
        return _t_rand_like(x) >= p

    def perm(self, n: int) -> torch.Tensor:
        return _t_randperm(n)

    def coin(self) -> torch.Tensor:
        return _t_rand(1)

    def sample(self, population: int, k: int) -> torch.Tensor:
        return torch.tensor(random.sample(range(population), k), dtype=torch.long)

    def neg_edges(self, fn):
        """``fn()`` runs the sampler (using ``self.sample``) and returns ``[2,K]``."""
        return fn()


class TapeRand(TorchRand):
    """TorchRand that also records every draw."""

    def __init__(self):
        self.tape: Tape = []

    def _rec(self, kind, t):
        self.tape.append((kind, t.clone()))
        return t

    def feat_keep(self, x, p):
        return self._rec("feat_keep", super().feat_keep(x, p))

    def edge_keep(self, num_edges, p):
        return self._rec("edge_keep", super().edge_keep(num_edges, p))

    def drop_keep(self, x, p):
        return self._rec("drop_keep", super().drop_keep(x, p))

    def perm(self, n):
        return self._rec("perm", super().perm(n))

    def coin(self):
        return self._rec("coin", super().coin())

    def neg_edges(self, fn):
        return self._rec("neg_edges", fn())


class ReplayRand:
    """Replays a tape in order (CPU side; the HIP side has its own in ``biomedkg_amd.rand``)."""

    def __init__(self, tape: Tape):
        self.tape = list(tape)
        self.pos = 0

    def _next(self, kind):
        k, t = self.tape[self.pos]
        if k != kind:
            raise RuntimeError(f"tape mismatch at {self.pos}: have {k!r}, asked {kind!r}")
        self.pos += 1
        return t

    def feat_keep(self, x, p):
        return self._next("feat_keep")

    def edge_keep(self, num_edges, p):
        return self._next("edge_keep")

    def drop_keep(self, x, p):
        return self._next("drop_keep")

    def perm(self, n):
        return self._next("perm")

    def coin(self):
        return self._next("coin")

    def sample(self, population, k):
        raise RuntimeError("ReplayRand replays whole neg_edges results, not raw samples")

    def neg_edges(self, fn):
        return self._next("neg_edges")


_current = TorchRand()


def current():
    return _current


class use:
    """``with oracle.rand.use(provider): ...`` installs a provider for the oracle ops."""

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


# ---- the library's in-kernel keep decisions of ReDAF's dropout, restated (test infrastructure: pins the mask definition)
def redaf_keep_mask(rows: int, d: int, p: float, seed: int, stream_id: int, row0: int = 0) -> torch.Tensor:
    """``biomedkg_amd/csrc/common.h hash_keep4`` for the ``[rows, d]`` slice starting at global row ``row0`` of a ReDAF input
    (reference ``biomedkg/utils/fusion.py:84``: ``nn.Dropout`` on the scaled transform -- WHICH elements are dropped is the
    library's own counter-based choice; this restates it).  Element ``e = (row0 + row) * d + col``: the four elements of a
    group ``q = e >> 2`` take the 16-bit fields of ``a = mix(mix(q ^ k0) + k1)`` and ``b = mix(a ^ rotl(q, 13) ^ c)`` (mix =
    murmur3's 32-bit finaliser), kept iff ``field << 16 >= uint32(p * 2^32)``."""
    import numpy as np
    assert d % 4 == 0
    m32 = np.uint64(0xFFFFFFFF)

    def mix(x):
        x = x & m32
        x ^= x >> np.uint64(16); x = (x * np.uint64(0x85EBCA6B)) & m32
        x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE35)) & m32
        x ^= x >> np.uint64(16)
        return x

    e = (np.uint64(row0) + np.arange(rows, dtype=np.uint64)[:, None]) * np.uint64(d) + np.arange(0, d, 4, dtype=np.uint64)[None, :]
    q, qh = (e >> np.uint64(2)) & m32, (e >> np.uint64(34)) & m32
    k0 = np.uint64((seed & 0xFFFFFFFF) ^ ((stream_id * 0x9E3779B1) & 0xFFFFFFFF))
    k1 = (np.uint64((seed >> 32) & 0xFFFFFFFF) + ((qh * np.uint64(0x85EBCA6B)) & m32)) & m32
    a = mix((mix(q ^ k0) + k1) & m32)
    rot = ((q << np.uint64(13)) | (q >> np.uint64(19))) & m32
    b = mix(a ^ rot ^ np.uint64(0x68E31DA4))
    thresh = np.uint64(min(int(np.float32(p) * np.float32(4294967296.0)), 4294967295))
    keep = np.empty((rows, d), dtype=bool)
    keep[:, 0::4] = ((a << np.uint64(16)) & m32) >= thresh
    keep[:, 1::4] = (a & np.uint64(0xFFFF0000)) >= thresh
    keep[:, 2::4] = ((b << np.uint64(16)) & m32) >= thresh
    keep[:, 3::4] = (b & np.uint64(0xFFFF0000)) >= thresh
    return torch.from_numpy(keep)
