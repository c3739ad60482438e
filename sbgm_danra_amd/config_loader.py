"""YAML run configuration with the reference's schema (reference sbgm/config/*.yaml, loader sbgm/utils.py:1626-1640).

The reference loads its YAML through OmegaConf with a custom `${env:VAR}` resolver and then reads it both as a dict
(`cfg['training']['seed']`) and by attribute (`cfg.evaluation.seed`).  OmegaConf is not a dependency here: this is a
small PyYAML (SafeLoader) loader giving the same two access styles and the same `${env:VAR}` interpolation
(unset variables resolve to None when the whole value is the reference, to "" inside a longer string).
"""
from __future__ import annotations

import os
import re

import yaml

_ENV = re.compile(r"\$\{env:([A-Za-z_][A-Za-z0-9_]*)\}")


class Config(dict):
    """dict with attribute access, recursively"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return super().get(k, default)


def _wrap(o):
    if isinstance(o, dict):
        return Config({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    if isinstance(o, str):
        m = _ENV.fullmatch(o)
        if m:
            v = os.environ.get(m.group(1))
            if v is not None and re.fullmatch(r"-?\d+", v):
                return int(v)
            return v
        return _ENV.sub(lambda mm: os.environ.get(mm.group(1), ""), o)
    return o


def to_config(d) -> Config:
    return _wrap(d)


def load_config(config_path: str) -> Config:
    if not os.path.exists(config_path):
        raise FileNotFoundError(f"Config file does not exist: {config_path}")
    with open(config_path) as f:
        return _wrap(yaml.load(f, Loader=yaml.SafeLoader) or {})
