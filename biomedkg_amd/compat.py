"""LightningModule surface for the task modules.

If ``lightning`` is importable the real ``LightningModule`` is used, so ``train_gcl.py`` /
``train_kge.py`` run under the real Trainer unchanged.  It is not installable in this image, so a
minimal stand-in provides exactly what the reference's modules and callers touch
(``save_hyperparameters`` -- callable before ``Module.__init__`` as ``gcl_module.py:115`` does --
``hparams``, ``device``, ``log`` / ``log_dict``, ``trainer``, ``load_from_checkpoint`` on the
Lightning checkpoint layout ``{"state_dict", "hyper_parameters"}``).
"""
from __future__ import annotations

import inspect
import types

import torch

try:  # pragma: no cover - not available in the build image
    from lightning import LightningModule  # type: ignore
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class _HParams(dict):
        __getattr__ = dict.__getitem__

    class LightningModule(torch.nn.Module):
        trainer = None

        def save_hyperparameters(self, *args, ignore=None, **kwargs):
            frame = inspect.currentframe().f_back
            info = inspect.getargvalues(frame)
            hp = {k: info.locals[k] for k in info.args if k != "self"}
            for k in ([ignore] if isinstance(ignore, str) else (ignore or [])):
                hp.pop(k, None)
            store = self.__dict__.setdefault("_hparams", _HParams())
            store.update(hp)

        @property
        def hparams(self):
            return self.__dict__.setdefault("_hparams", _HParams())

        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def log(self, name, value, **kwargs):
            logged = self.__dict__.setdefault("logged_metrics", {})
            logged[name] = value.detach() if isinstance(value, torch.Tensor) else value
            tr = self.__dict__.get("trainer") or type(self).trainer
            if tr is not None and hasattr(tr, "_log"):
                tr._log(name, logged[name], **kwargs)

        def log_dict(self, metrics, **kwargs):
            for k, v in metrics.items():
                self.log(k, v, **kwargs)

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **overrides):
            ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(overrides)
            accepted = inspect.signature(cls.__init__).parameters
            model = cls(**{k: v for k, v in hp.items() if k in accepted})
            state = ckpt["state_dict"]
            own = set(model.state_dict().keys())
            # torchmetrics states saved by the reference's modules are not parameters of this build
            model.load_state_dict({k: v for k, v in state.items() if k in own or strict and not _is_metric_key(k)},
                                  strict=strict)
            return model


def _is_metric_key(key: str) -> bool:
    return key.startswith(("valid_metrics.", "test_metrics.", "edge_wise_pre_"))


def namespace(**kw):
    return types.SimpleNamespace(**kw)
