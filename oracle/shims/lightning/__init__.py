"""Shim LightningModule: just enough of the API the reference's modules touch."""
import inspect
import types

import torch


class LightningModule(torch.nn.Module):
    def save_hyperparameters(self, *args, ignore=None, **kwargs):
        # The GCL subclasses call this BEFORE Module.__init__ (gcl_module.py:115), so write
        # straight into __dict__.
        frame = inspect.currentframe().f_back
        names = inspect.getargvalues(frame)
        hp = {k: names.locals[k] for k in names.args if k != "self"}
        for k in ignore or []:
            hp.pop(k, None)
        prev = self.__dict__.get("_hparams", {})
        prev.update(hp)
        self.__dict__["_hparams"] = prev

    @property
    def hparams(self):
        return types.SimpleNamespace(**self.__dict__.get("_hparams", {}))

    @property
    def device(self):
        return next(self.parameters()).device

    def log(self, name, value, **kwargs):
        self.__dict__.setdefault("_logged", {})[name] = value

    def log_dict(self, d, **kwargs):
        self.__dict__.setdefault("_logged", {}).update(d)


class LightningDataModule:
    """Shim: ``save_hyperparameters`` is all the reference's data modules call (``biomedkg/data_module.py:40-41``)."""

    def save_hyperparameters(self, *args, **kwargs):
        pass
