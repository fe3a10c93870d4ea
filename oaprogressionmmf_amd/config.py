"""Minimal stand-in for OmegaConf's DictConfig: the reference reads its config both as `config["k"]`
and `config.k` (koafusion/models/_xrNmrMcP.py:36,259), so any substitute needs item + attribute access.
Real DictConfig objects work too (nothing here depends on this class)."""
from pathlib import Path

import yaml


class ConfigDict(dict):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(e) for e in v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = self._wrap(v)

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))


CONF_DIR = Path(__file__).resolve().parent / "run" / "conf"


def load_model_config(name, **overrides):
    """conf/model/<name>.yaml (+ dotted overrides, e.g. **{"fe.xr.arch": "resnext50_32x4d"})"""
    cfg = ConfigDict(yaml.safe_load((CONF_DIR / "model" / f"{name}.yaml").read_text()))
    for k, v in overrides.items():
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg
