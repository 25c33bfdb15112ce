"""Configuration: the keys of reference configs/soc.yaml:6-43, Hydra-style `a.b=value` overrides.

The reference needs hydra + omegaconf + the submitit launcher plugin (configs/soc.yaml:1-2), none
of which exist in this image.  `load_config` reads our `configs/soc.yaml` with PyYAML and applies
command-line overrides; `main.py` uses real Hydra instead when it is importable.
"""
import ast
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_YAML = os.path.join(os.path.dirname(_HERE), "configs", "soc.yaml")


class Config(dict):
    """dict with attribute access (what the code needs from omegaconf.DictConfig)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return Config({k: Config.wrap(v) for k, v in obj.items()})
        return obj


def _parse_value(text):
    low = text.strip()
    if low.lower() in ("true", "false"):
        return low.lower() == "true"
    if low.lower() in ("none", "null"):
        return None
    try:
        return ast.literal_eval(low)
    except (ValueError, SyntaxError):
        return low.strip("'\"")


def apply_overrides(cfg, overrides):
    for item in overrides:
        if "=" not in item:
            raise ValueError(f"override {item!r} is not of the form key.path=value")
        key, val = item.split("=", 1)
        node = cfg
        parts = key.lstrip("+").split(".")
        for p in parts[:-1]:
            if p not in node:
                node[p] = Config()
            node = node[p]
        node[parts[-1]] = _parse_value(val)
    return cfg


def load_config(overrides=(), path=DEFAULT_YAML):
    with open(path) as f:
        raw = yaml.safe_load(f)
    raw.pop("defaults", None)
    raw.pop("hydra", None)
    cfg = Config.wrap(raw)
    # yaml reads 1e-4 as a string; normalise numerics
    for sec in ("method", "optim"):
        for k, v in list(cfg[sec].items()):
            if isinstance(v, str):
                cfg[sec][k] = _parse_value(v)
    return apply_overrides(cfg, overrides)
