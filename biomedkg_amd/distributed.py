"""Multi-GPU glue: one process per GPU, ``torch.distributed`` over RCCL (backend "nccl" on ROCm).

The reference's only multi-GPU mode is Lightning's implicit DDP over mini-batches
(``configs/{gcl,kge,dpi}.yaml:11``, ``train_gcl.py:63-68``): every rank steps on its own
batch / graph and the gradients of ``self.model`` are averaged.  That is what this module provides,
sized for xGMI: the R-GCN (R=30, d=256) gradient is 32.5 MB and the GCN+head 1.8 MB, so ALL gradients
go out as ONE flat bucket per step (a ring all-reduce over 7 x ~153 GB/s links is per-link bound:
2*(7/8)*32.5 MB / 153 GB/s ~ 0.37 ms; splitting it would only add launch latency).  No collective is on
the data path of a single-GPU step.  Node-partitioned full-graph training (destination-node shards, all-gather of
the layer tables / reduce-scatter of their gradients) lives in ``biomedkg_amd/partition.py`` and uses
the bucket below for the replicated weights.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*.
    Returns (rank, world, local_rank).  No-op for WORLD_SIZE == 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


class _Done:
    @staticmethod
    def wait():
        return True


class GradBucket:
    """Averages the gradients of ``params`` across ranks with one flat all-reduce per call."""

    def __init__(self, params, world=None, group=None, simulate=False):
        """``group``: the process group the all-reduce runs on (a bucket reduced on a stream of its own while another
        communicator's collectives are in flight -- bench.py's graphed partitioned steps -- needs a group of its own).
        ``simulate``: one process standing in for a rank of ``world`` (partition.Comm(simulate=True)): pack, scale and
        write-back as live, the all-reduce itself replaced by a local round trip of the bucket's bytes."""
        self.simulate = bool(simulate)
        self.params = [p for p in params if p.requires_grad]
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.group = group
        self._flat = None
        # (BKG_PARTITION_FORCE_COLLECTIVES=1: a 1-rank group still issues its all-reduce -- functional check of the calls)
        self.live = (self.world > 1 or (os.environ.get("BKG_PARTITION_FORCE_COLLECTIVES") == "1" and dist.is_initialized())
                     or self.simulate)

    def pack(self):
        """The gradients copied into the flat bucket (``views``: the bucket seen through each parameter's shape)."""
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        grads = [p.grad for p in self.params]
        n = sum(g.numel() for g in grads)
        if self._flat is None or self._flat.numel() != n or self._flat.device != grads[0].device:
            self._flat = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
            self._views, off = [], 0
            for g in grads:                       # persistent views of the bucket, one per parameter
                self._views.append(self._flat[off:off + g.numel()].view(g.shape))
                off += g.numel()
        torch._foreach_copy_(self._views, grads)  # one fused pack instead of a copy_ launch per parameter

    def begin(self):
        """Pack the gradients into the flat bucket and start the all-reduce (asynchronously: the collective runs on
        the backend's own stream / thread).  ``finish`` must be called before the gradients are used."""
        if not self.live:
            return
        self.pack()
        if self.simulate:
            self._flat.copy_(self._flat.clone())
            self._work = _Done
            return
        self._work = dist.all_reduce(self._flat, async_op=True, group=self.group)

    def finish(self):
        """Wait for the collective started by ``begin`` and write the averaged gradients back."""
        if not self.live or getattr(self, "_work", None) is None:
            return
        self._work.wait()
        self._work = None
        self._flat.div_(self.world)
        torch._foreach_copy_([p.grad for p in self.params], self._views)

    def all_reduce(self):
        self.begin()
        self.finish()

    def all_reduce_packed(self):
        """The all-reduce of a bucket that ``pack`` filled (synchronous on the current stream); ``views`` hold the sums."""
        if self.simulate:
            self._flat.copy_(self._flat.clone())
        elif self.live:
            dist.all_reduce(self._flat, group=self.group)


def broadcast_parameters(module, src=0):
    """Make every rank start from rank ``src``'s weights (what DDP does at construction)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_object(obj, src=0):
    """Rank ``src``'s python object on every rank (paths, small metadata)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def mean_over_ranks(values, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return (t / dist.get_world_size()).tolist()


def any_over_ranks(flag: bool, device) -> bool:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
