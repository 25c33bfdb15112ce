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


def split_sweep(value):
    """Hydra's comma-separated sweep syntax: `a,b,c` -> ['a', 'b', 'c'], commas inside brackets / quotes kept
    (`arch.hdims_M=[64,64]` is ONE value)."""
    out, depth, quote, cur = [], 0, None, []
    for ch in value:
        if quote:
            cur.append(ch)
            if ch == quote:
                quote = None
        elif ch in "'\"":
            quote = ch
            cur.append(ch)
        elif ch in "[({":
            depth += 1
            cur.append(ch)
        elif ch in "])}":
            depth -= 1
            cur.append(ch)
        elif ch == "," and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    out.append("".join(cur))
    return out


def expand_multirun(argv):
    """The reference's README command lines (README.md:15-60) are Hydra MULTIRUNS: `method.algorithm='SOCM','SOCM_const_M',... -m`.
    Returns (list of override lists, multirun flag): the cartesian product of the comma lists, in Hydra's job order (the
    LAST swept key varies fastest), one job per combination.  A comma list without -m / --multirun is the error it is
    under Hydra."""
    import itertools
    multirun = any(a in ("-m", "--multirun") for a in argv)
    overrides = [a for a in argv if "=" in a and not a.startswith("-")]
    keys, choices = [], []
    for item in overrides:
        key, val = item.split("=", 1)
        keys.append(key)
        choices.append(split_sweep(val))
    swept = [k for k, c in zip(keys, choices) if len(c) > 1]
    if swept and not multirun:
        raise ValueError(f"override(s) {swept} hold a comma-separated list: add -m / --multirun to sweep them "
                         "(one run per value, as Hydra does), or quote the value")
    jobs = [[f"{k}={v}" for k, v in zip(keys, combo)] for combo in itertools.product(*choices)]
    return jobs, multirun


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
