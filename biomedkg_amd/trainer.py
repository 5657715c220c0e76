"""Minimal stand-in for the Lightning ``Trainer`` behaviours the reference's entry points rely on
(``train_gcl.py:63-122``, ``train_kge.py:44-98``; SURVEY.md 8b "Trainer behaviours to reproduce"):
``seed_everything``, ``max_epochs``, ``check_val_every_n_epoch``, ``gradient_clip_val`` (global L2
norm over the optimiser's parameters), ``fast_dev_run``, ``ModelCheckpoint(monitor, mode, save_top_k,
save_last)``, ``EarlyStopping(monitor, patience)``, ``estimated_stepping_batches``, the per-EPOCH
``lr_scheduler.step()`` Lightning applies to a bare scheduler, ``test(ckpt_path="best")``
and -- for ``devices`` > 1 under torch.distributed.run -- DDP-style data parallelism over RCCL: every rank draws its
OWN batches (per-rank seeds; loaders that define ``set_rank`` shard their seed batches), gradients are averaged through
one flat bucket, the monitored metrics are averaged and ``should_stop`` is OR-ed over the ranks before the callbacks
act on them (so no rank can leave the loop while the others wait in a collective), and checkpoint paths are rank 0's.
Used only when ``lightning`` is not importable.
"""
from __future__ import annotations

import json
import os
import random
import time

import numpy as np
import torch

from . import distributed as D
from .optim import clip_adam_step


def seed_everything(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PL_GLOBAL_SEED"] = str(seed)
    return seed


class ModelCheckpoint:
    def __init__(self, dirpath, monitor="val_loss", save_top_k=1, mode="min", save_last=False):
        self.dirpath, self.monitor, self.save_top_k, self.mode, self.save_last = dirpath, monitor, save_top_k, mode, save_last
        self.best = []            # [(score, path)]
        self.best_model_path = ""

    def _better(self, a, b):
        return a < b if self.mode == "min" else a > b

    def on_validation_end(self, trainer, module, metrics):
        if self.monitor not in metrics or trainer.fast_dev_run:
            return
        score = float(metrics[self.monitor])
        os.makedirs(self.dirpath, exist_ok=True)
        path = os.path.join(self.dirpath, f"epoch={trainer.current_epoch}-step={trainer.global_step}.ckpt")
        self.best.append((score, path))
        self.best.sort(key=lambda t: t[0], reverse=self.mode != "min")
        trainer.save_checkpoint(module, path)
        for _, old in self.best[self.save_top_k:]:
            if trainer.rank == 0 and os.path.exists(old):
                os.remove(old)
        self.best = self.best[: self.save_top_k]
        self.best_model_path = self.best[0][1]
        if self.save_last:
            trainer.save_checkpoint(module, os.path.join(self.dirpath, "last.ckpt"))


class EarlyStopping:
    def __init__(self, monitor="val_loss", mode="min", patience=3):
        self.monitor, self.mode, self.patience = monitor, mode, patience
        self.best, self.wait = None, 0

    def on_validation_end(self, trainer, module, metrics):
        if self.monitor not in metrics:
            return
        score = float(metrics[self.monitor])
        if self.best is None or (score < self.best if self.mode == "min" else score > self.best):
            self.best, self.wait = score, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                trainer.should_stop = True


class Trainer:
    def __init__(self, max_epochs=1, check_val_every_n_epoch=1, gradient_clip_val=None, fast_dev_run=False,
                 callbacks=None, devices=None, accelerator="auto", log_every_n_steps=10, deterministic=False,
                 enable_checkpointing=True, default_root_dir=None, logger=None, **unused):
        self.max_epochs, self.val_every = max_epochs, check_val_every_n_epoch
        self.clip, self.fast_dev_run = gradient_clip_val, fast_dev_run
        self.callbacks = list(callbacks or [])
        self.log_every, self.root = log_every_n_steps, default_root_dir
        self.rank, self.world, self.local = D.init_from_env()
        self.device = torch.device("cuda", self.local) if torch.cuda.is_available() else torch.device("cpu")
        if self.world > 1:
            # the entry points stamp their checkpoint directory with time.time() on every rank: use rank 0's everywhere
            dirs = [self.root] + [getattr(cb, "dirpath", None) for cb in self.callbacks]
            dirs = D.broadcast_object(dirs, src=0)
            self.root = dirs[0]
            for cb, d in zip(self.callbacks, dirs[1:]):
                if d is not None:
                    cb.dirpath = d
        if deterministic and torch.cuda.is_available():
            from . import ops
            ops.set_deterministic(True)          # fixed-order weight-gradient sums (Lightning: deterministic=True)
        self.current_epoch = self.global_step = 0
        self.should_stop = False
        self.estimated_stepping_batches = 0
        self.callback_metrics = {}
        self._epoch_logs = {}
        self.history = []

    # ---- logging hook used by compat.LightningModule.log
    def _log(self, name, value, **kw):
        v = float(value.item() if isinstance(value, torch.Tensor) else value)
        self._epoch_logs.setdefault(name, []).append(v)

    def _flush_logs(self):
        out = {k: sum(v) / len(v) for k, v in self._epoch_logs.items()}
        self._epoch_logs = {}
        if self.world > 1 and out:               # every rank logs the same keys: average them (Lightning's sync_dist)
            keys = sorted(out)
            vals = D.mean_over_ranks([out[k] for k in keys], self.device)
            out = dict(zip(keys, vals))
        self.callback_metrics.update(out)
        return out

    def _try_graph(self, model, loader, opt):
        """Whole-graph loaders hand out the SAME batch every step: capture the step once into a hipGraph and replay it
        (biomedkg_amd/graphs.py) when the module's step has no host-side draws (GRACE); otherwise stay eager.  Off with
        BKG_HIPGRAPH=0, for multi-rank runs (the gradient all-reduce is not part of the capture) and under fast_dev_run."""
        batch = getattr(loader, "batch", None)
        if (self.world > 1 or self.fast_dev_run or os.environ.get("BKG_HIPGRAPH", "1") == "0"
                or self.device.type != "cuda"):
            return None
        from .samplers import SaintBatchLoader
        if isinstance(loader, SaintBatchLoader) and getattr(getattr(model, "model", None), "decoder", None) is not None \
                and getattr(model.model.decoder, "kind", None) is not None \
                and isinstance(getattr(model, "neg_ratio", None), int):      # (dpi.yaml's default is the STRING "none")
            # the reference's KGE regime (GraphSAINT batches): padded sync-free batches, one graph per edge bucket
            from .sampled import SaintGraphedKge
            sm = loader.sampler
            loop = SaintGraphedKge(model, loader.x, sm.edge_index, loader.edge_type, sm.batch_size, sm.walk_length, opt,
                                   clip=self.clip, seed=sm.seed)
            step = loop.step
            step_fn = lambda: step()      # noqa: E731
            step_fn.loop, step_fn.owns_batches, step_fn.warmup_steps = loop, True, 0
            return step_fn
        from .samplers import NeighborBatchLoader
        from .model.gcl import DGI, GGD, GRACE
        if isinstance(loader, NeighborBatchLoader) and isinstance(getattr(model, "model", None), (GGD, DGI, GRACE)) \
                and loader.sampler.one_read_ok() and loader.x.size(0) > 0:
            # the reference's GCL regime (NeighborLoader batches): padded sync-free batches, one graph per edge bucket and
            # augmentation branch (biomedkg_amd/sampled_gcl.py)
            from .sampled_gcl import NeighborGraphedGcl
            sm = loader.sampler
            loop = NeighborGraphedGcl(model, loader.x, sm.edge_index, loader.batch_size, sm.num_neighbors, opt,
                                      clip=self.clip, seed=sm.seed, shuffle=loader.shuffle)
            step = loop.step
            step_fn = lambda: step()      # noqa: E731
            step_fn.loop, step_fn.owns_batches, step_fn.warmup_steps = loop, True, 0
            return step_fn
        if batch is None:
            return None
        from . import graphs
        import copy
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        opt_state = copy.deepcopy(opt.state_dict())              # param_groups too: make_capturable rewrites lr / capturable
        rng = torch.cuda.get_rng_state(self.device)
        try:
            # rewind: the warm-up passes are undone before the capture, the run has exactly epochs x steps optimizer steps
            step = graphs.GraphedStep(model, self._move(batch), opt, clip=self.clip, warmup=2, rewind=True)
        except (graphs.CaptureRefused, RuntimeError) as why:      # refused up front, or an op on the path cannot be captured
            torch.cuda.synchronize()
            model.load_state_dict(state)
            opt.load_state_dict(opt_state)                        # Adam's moments / step counters of the warm-up steps go too
            torch.cuda.set_rng_state(rng, self.device)
            if self.rank == 0:
                print(f"[trainer] eager steps: {why}", flush=True)
            return None
        self.global_step += step.warmup_steps       # the capture's eager warm-up steps are real optimizer steps
        return step

    def _seed_rank(self, loaders):
        """Per-rank randomness: without it every rank would draw the same batches, masks and negatives and N GPUs
        would do one GPU's work (Lightning shards through DistributedSampler; the device samplers through set_rank)."""
        if self.world == 1:
            return
        base = int(os.environ.get("PL_GLOBAL_SEED", torch.initial_seed())) & 0x7FFFFFFF
        random.seed(base + self.rank)
        np.random.seed((base + self.rank) % (1 << 32))
        torch.manual_seed(base + self.rank)
        from . import rand as R
        R.reseed_default(base + self.rank)
        for ld in loaders:
            if ld is not None and hasattr(ld, "set_rank"):
                ld.set_rank(self.rank, self.world)

    def save_checkpoint(self, module, path):
        if self.rank != 0:
            return
        hp = {k: v for k, v in dict(getattr(module, "hparams", {})).items() if not isinstance(v, torch.nn.Module)}
        torch.save({"state_dict": module.state_dict(), "hyper_parameters": hp, "epoch": self.current_epoch,
                    "global_step": self.global_step}, path)

    def _move(self, batch):
        """``batch.to(device)`` -- once per batch OBJECT: a whole-graph loader hands out the same host batch every step
        (reference data_module.py:89-145 with a full-batch loader; Lightning would copy it to the device every step), and a
        fresh device copy each step is 170 MB over PCIe plus every graph structure rebuilt on new tensors (a full-size KGE
        step: 16.7 ms instead of 5).  The copy is kept while the loader keeps yielding the same object with unchanged
        tensors (their ``_version`` counters)."""
        if not hasattr(batch, "to"):
            return batch
        tensors = [v for v in vars(batch).values() if isinstance(v, torch.Tensor)] if hasattr(batch, "__dict__") else []
        key = tuple((id(t), t._version) for t in tensors)
        kept = self.__dict__.setdefault("_moved", {})             # id(batch) -> (key, the batch itself, its device copy)
        hit = kept.get(id(batch))
        if hit is not None and hit[0] == key and hit[1] is batch:
            return hit[2]
        moved = batch.to(self.device)
        if tensors:
            kept[id(batch)] = (key, batch, moved)
            while len(kept) > 4:                                  # (train / val / test loaders' batches)
                kept.pop(next(iter(kept)))
        return moved

    def _limit(self, loader):
        for i, batch in enumerate(loader):
            if self.fast_dev_run and i >= 1:
                break
            yield i, self._move(batch)

    def fit(self, model, train_dataloaders=None, val_dataloaders=None):
        model.trainer = self
        model.to(self.device)
        D.broadcast_parameters(model)
        self._seed_rank([train_dataloaders, val_dataloaders])
        steps_per_epoch = len(train_dataloaders) if hasattr(train_dataloaders, "__len__") else 1
        epochs = 1 if self.fast_dev_run else self.max_epochs
        self.estimated_stepping_batches = 1 if self.fast_dev_run else steps_per_epoch * epochs
        conf = model.configure_optimizers()
        opt, sched = conf["optimizer"], conf.get("lr_scheduler")
        params = [p for g in opt.param_groups for p in g["params"]]
        bucket = D.GradBucket(params, self.world)
        model.train()                       # (a fit() after test() would otherwise capture an eval-mode step)
        graphed = self._try_graph(model, train_dataloaders, opt)
        for epoch in range(epochs):
            self.current_epoch = epoch
            model.train()
            t0 = time.time()
            batches = self._limit(train_dataloaders)
            if graphed is not None and getattr(graphed, "owns_batches", False):
                # the graphed loop samples for itself (fast_dev_run never gets here: _try_graph declines it)
                batches = ((i, None) for i in range(len(train_dataloaders)))
            for i, batch in batches:
                if graphed is not None:            # one hipGraph replay = zero_grad + step + backward + clip + Adam
                    loss = graphed()
                    if self.global_step % max(1, self.log_every) == 0:
                        self._log("train_loss", loss)
                    self.global_step += 1
                    continue
                opt.zero_grad(set_to_none=True)
                loss = model.training_step(batch)
                loss.backward()
                bucket.all_reduce()
                clip_adam_step(opt, self.clip)     # clip_grad_norm_ + opt.step(): two launches (biomedkg_amd/optim.py)
                self.global_step += 1
            if sched is not None:
                sched.step()                      # Lightning's default interval for a bare scheduler: "epoch"
            logs = self._flush_logs()
            logs.update(epoch=epoch, seconds=time.time() - t0)
            if val_dataloaders is not None and (self.fast_dev_run or (epoch + 1) % self.val_every == 0):
                logs.update(self.validate(model, val_dataloaders))
                for cb in self.callbacks:
                    cb.on_validation_end(self, model, self.callback_metrics)
                self.should_stop = D.any_over_ranks(self.should_stop, self.device)
            self.history.append(logs)
            if self.rank == 0:
                print(json.dumps({k: (round(v, 6) if isinstance(v, float) else v) for k, v in logs.items()}), flush=True)
            if self.should_stop:
                break
        if graphed is not None and hasattr(getattr(graphed, "loop", None), "close"):
            graphed.loop.close()              # the sampler runs a batch ahead on its own stream
        return self

    @torch.no_grad()
    def validate(self, model, loader):
        model.eval()
        for i, batch in self._limit(loader):
            model.validation_step(batch, i)
        if hasattr(model, "on_validation_epoch_end"):
            model.on_validation_epoch_end()
        model.train()
        return self._flush_logs()

    @torch.no_grad()
    def test(self, model=None, dataloaders=None, ckpt_path=None):
        model.trainer = self
        model.to(self.device)
        if ckpt_path == "best":
            best = next((cb.best_model_path for cb in self.callbacks if isinstance(cb, ModelCheckpoint)), "")
            if self.world > 1:                   # rank 0 wrote the file: wait for it, then every rank loads the same one
                D.barrier()
                best = D.broadcast_object(best, src=0)
            if best and os.path.exists(best):
                model.load_state_dict(torch.load(best, map_location=self.device, weights_only=False)["state_dict"])
        model.eval()
        for i, batch in self._limit(dataloaders):
            model.test_step(batch, i)
        if hasattr(model, "on_test_epoch_end"):
            model.on_test_epoch_end()
        out = self._flush_logs()
        if self.rank == 0:
            print(json.dumps({"test": {k: round(v, 6) for k, v in out.items()}}), flush=True)
        return [out]
