"""Minimal stand-in for `mmengine.Config.fromfile` (reference main.py:338): a Python file whose
top-level names become config entries, with attribute access and `.get`."""
from __future__ import annotations

import os
import runpy


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


class Config(ConfigDict):
    @classmethod
    def fromfile(cls, path: str) -> "Config":
        if not os.path.isfile(path):
            raise FileNotFoundError(path)
        ns = runpy.run_path(path)
        return cls({k: _wrap(v) for k, v in ns.items() if not k.startswith("_") and not callable(v)
                    and not isinstance(v, type(os))})
