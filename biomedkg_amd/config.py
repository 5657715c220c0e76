"""A small Hydra-compatible composer for the ``configs/`` tree (Hydra / omegaconf are not installable
in this image).  Supports exactly what the reference's entry points and launch scripts use
(SURVEY.md 5.6): ``defaults:`` lists with ``_self_`` and ``group: name`` entries (nested, e.g.
``model/gcl.yaml`` -> ``base``), ``a.b=value`` dot overrides with YAML-typed values (``devices=[0]``),
group overrides (``data=synthetic``), ``_target_`` instantiation and ``**cfg.model`` expansion.
If hydra is importable the real ``@hydra.main`` can be used instead; the YAML files are the same.
"""
from __future__ import annotations

import importlib
import os

import yaml


class Config(dict):
    """dict with attribute access (enough of DictConfig for ``cfg.model.in_dim`` / ``**cfg.model``)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value


def _wrap(obj):
    if isinstance(obj, dict):
        return Config({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    return obj


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load(path):
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _load_group_file(config_dir, group, name):
    """A file inside a group; its own ``defaults`` are siblings in the same group directory."""
    body = _load(os.path.join(config_dir, group, name + ".yaml"))
    out = {}
    for d in body.pop("defaults", []):
        if d == "_self_":
            _merge(out, body)
            body = {}
        else:
            _merge(out, _load_group_file(config_dir, group, d))
    return _merge(out, body)


def compose(config_dir, config_name, overrides=()):
    root = _load(os.path.join(config_dir, config_name + ".yaml"))
    defaults = root.pop("defaults", ["_self_"])
    group_choice, dotted = {}, []
    for ov in overrides:
        key, _, val = ov.partition("=")
        if "." not in key and os.path.isdir(os.path.join(config_dir, key)):
            group_choice[key] = val
        else:
            dotted.append((key, yaml.safe_load(val)))
    cfg = {}
    for d in defaults:
        if d == "_self_":
            _merge(cfg, root)
        else:
            (group, name), = d.items()
            cfg[group] = _merge(cfg.get(group, {}), _load_group_file(config_dir, group, group_choice.get(group, name)))
    for key, val in dotted:
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = val
    return _wrap(cfg)


def instantiate(cfg, **extra):
    """``hydra.utils.instantiate`` for a flat ``_target_`` node."""
    kwargs = {k: v for k, v in cfg.items() if k != "_target_"}
    kwargs.update(extra)
    module, _, name = cfg["_target_"].rpartition(".")
    # the reference's configs name its own package (configs/data/primekg.yaml: ``biomedkg.data_module.PrimeKGModule``): this
    # build is the drop-in for those classes, so the verbatim target resolves to the class of the same name here
    if module == "biomedkg" or module.startswith("biomedkg."):
        module = "biomedkg_amd" + module[len("biomedkg"):]
    return getattr(importlib.import_module(module), name)(**kwargs)
