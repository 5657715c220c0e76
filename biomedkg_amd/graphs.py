"""hipGraph capture of a whole training step for the launch-bound regime (small full-graph batches: BASELINE
configs[0-1], the 5k-node / 50k-edge GCL graph, where a GRACE step is ~300 launches of a few microseconds each).

``GraphedStep(module, batch, optimizer, clip)`` records ONE step -- replay-epoch bump, ``training_step`` (fusion,
augmentation, encoder passes, projector, NT-Xent), ``backward``, gradient clipping, Adam -- into a hipGraph
(``torch.cuda.CUDAGraph`` on ROCm: every C-ABI launch of ``libbiomedkg_hip.so`` goes to torch's current stream, which is
the capturing stream) and replays it: one host call per step instead of one per kernel.

What makes the step capturable, and what is required of it:

* static shapes and buffers: the SAME batch object every step (the whole-graph loaders of ``data_module.py``); graph
  structures (CSRs, gcn_norm work buffers) are built during the eager warm-up and only re-weighted inside the step;
* no host synchronisation inside ``training_step`` (GRACE: feature / edge masks are device draws, NT-Xent sizes its
  workspaces from shapes).  Steps that branch on the host (GGD's augmentation coin) or draw through a sort
  (``randperm``: DGI / GGD corruption) or count on the host (the KGE negative sampler) are refused with a clear error;
* fresh randomness per replay although kernel arguments are frozen: torch's own draws are graph-safe (the capture
  registers the generator); the in-kernel Philox streams of the library (layer dropout) add a DEVICE replay counter to
  their stream id (``bkg_set_rng_epoch``), which the captured step increments first thing;
* the optimizer state lives on the device (``capturable=True``; the learning rate is a device scalar so that the
  per-epoch scheduler can change it between replays without re-capturing).

Reference call sites replaced: the Lightning training loop body around ``BaseGCL.training_step``
(``biomedkg/gcl_module.py:60-64``; ``train_gcl.py:63-68`` builds the Trainer with ``gradient_clip_val=1.0``).
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops
from . import rand as R
from .optim import clip_adam_step


class CaptureRefused(RuntimeError):
    pass


class _GraphRand(R.DeviceRand):
    """DeviceRand whose host-side draws are refused: a captured step cannot branch on them."""

    def coin(self):
        raise CaptureRefused("this step draws a host-side coin (GGD augmentation): its control flow cannot be captured")

    def perm(self, n, device):
        raise CaptureRefused("this step draws a permutation through a sort (DGI / GGD corruption): not capturable")

    def neg_edges(self, edge_index, num_neg_samples):
        raise CaptureRefused("the KGE negative sampler counts its survivors on the host: not capturable")


def make_capturable(optimizer):
    """Adam with device-side step counters and a device learning rate (before its first step)."""
    for g in optimizer.param_groups:
        g["capturable"] = True
        if not isinstance(g["lr"], torch.Tensor):
            p0 = g["params"][0]
            g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=p0.device)
    return optimizer


class GraphedStep:
    def __init__(self, module, batch, optimizer, clip=1.0, warmup=3, seed=None, provider=None, rewind=False):
        """``rewind``: the eager warm-up steps (they build caches, Adam's state buffers and kernel plans) are undone
        before the capture -- parameters, optimizer state and torch's device generator go back IN PLACE to what they
        were -- so the first replay is the run's first optimizer step (``warmup_steps == 0``): a Trainer that captures
        follows the trajectory of one that does not.  Default: the warm-up steps are real steps of the run."""
        L.require_cuda(*[p for g in optimizer.param_groups for p in g["params"]][:1])
        self.rewind = bool(rewind)
        self.module, self.batch, self.optimizer, self.clip = module, batch, make_capturable(optimizer), clip
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        self.provider = provider if provider is not None else _GraphRand(seed)
        self.epoch = torch.zeros(1, dtype=torch.int32, device=self.params[0].device)
        self.graph, self.loss = None, None
        self.replays = 0
        self._capture(warmup)

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        self.epoch.add_(1)
        loss = self.module.training_step(self.batch)
        loss.backward()
        clip_adam_step(self.optimizer, self.clip)        # clip_grad_norm_ + step(): two launches (biomedkg_amd/optim.py)
        return loss

    def _capture(self, warmup):
        trainer = self.module.__dict__.pop("trainer", None)        # logging inside training_step would sync (.item())
        L.call("bkg_set_rng_epoch", L.ptr(self.epoch))
        try:
            with R.use(self.provider):
                if self.rewind:
                    torch.cuda.synchronize()
                    saved_p = [p.detach().clone() for p in self.params]
                    saved_s = {p: {k: v.detach().clone() for k, v in st.items() if isinstance(v, torch.Tensor)}
                               for p, st in self.optimizer.state.items()}
                    saved_rng = torch.cuda.get_rng_state(self.params[0].device)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                       # eager warm-up: caches, Adam state, kernel plans
                    for _ in range(max(1, warmup)):
                        self._eager()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                if self.rewind:
                    with torch.no_grad():
                        for p, q in zip(self.params, saved_p):
                            p.copy_(q)
                        for p, st in self.optimizer.state.items():      # in place: the capture freezes these addresses
                            for k, v in st.items():
                                if isinstance(v, torch.Tensor):
                                    was = saved_s.get(p, {}).get(k)
                                    v.copy_(was) if was is not None else v.zero_()   # Adam's fresh state is all zeros
                    torch.cuda.set_rng_state(saved_rng, self.params[0].device)
                    self.epoch.zero_()
                    if hasattr(self.provider, "counter"):
                        self.provider.counter = 0
                    torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                self.optimizer.zero_grad(set_to_none=True)
                with torch.cuda.graph(self.graph):
                    self.loss = self._eager()
        finally:
            # the captured launches carry the counter's address as a kernel argument: replays need no global state, and
            # nothing outside this object may keep launching with a pointer to a tensor this object owns
            L.call("bkg_set_rng_epoch", None)
            if trainer is not None:
                self.module.__dict__["trainer"] = trainer
        self.warmup_steps = 0 if self.rewind else max(1, warmup)    # (the capture pass records, it does not execute)

    def __call__(self):
        """One training step; returns the (static) loss tensor of the replay."""
        self.graph.replay()
        self.replays += 1
        ops.note_graph_replay()          # weights moved under an unchanged tensor version: per-version caches are stale
        return self.loss

    def close(self):
        """Kept for callers of the first version: the replay counter is only registered while capturing."""
